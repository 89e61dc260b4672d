// luma_exception.h -- error type thrown by the facade's initialize()/decode() paths, same name and
// interface as the reference's (include/luma/luma_exception.h:53-71 there): std::exception + what().
#ifndef LUMA_HIP_EXCEPTION_H
#define LUMA_HIP_EXCEPTION_H

#include <exception>
#include <string>

class LumaException : public std::exception {
public:
    explicit LumaException(const char *message) : m_what(message ? message : "") {}
    explicit LumaException(const std::string &message) : m_what(message) {}
    ~LumaException() throw() {}
    const char *what() const throw() { return m_what.c_str(); }

private:
    std::string m_what;
};

#endif
