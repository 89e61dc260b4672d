// luma_exception.h -- the error type the facade throws from initialize() / encode() / decode().
// Interface contract taken from the reference (include/luma/luma_exception.h:53-71 there): constructible from a
// C string, catchable as std::exception, what() returns the message.  Implemented on std::runtime_error.
#ifndef LUMA_HIP_EXCEPTION_H
#define LUMA_HIP_EXCEPTION_H

#include <stdexcept>
#include <string>

struct LumaException : std::runtime_error {
    explicit LumaException(const char *message) : std::runtime_error(message ? message : "unknown Luma HDRv error") {}
    explicit LumaException(const std::string &message) : std::runtime_error(message) {}
};

#endif
