// tools/luma_cli.h -- command-line option table shared by tools/lumaenc.cpp and tools/lumadec.cpp.
//
// Behavioural counterpart of the reference's option handling (lumaenc.cpp:106-179, lumadec.cpp:71-90, the ArgParser they
// use): the same option names and short names, defaults, value ranges / value sets and error situations (unknown
// option, missing value, out-of-range or invalid value, missing required option; -h / --help prints the option list and
// makes the program exit with status 1).  Own implementation: one table of typed option descriptors.
#ifndef LUMA_HIP_CLI_H
#define LUMA_HIP_CLI_H

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace lumacli {

// what the reference reports as "<tool> input error: ..."
struct UsageError : std::runtime_error {
    explicit UsageError(const std::string &m) : std::runtime_error(m) {}
};

class Options {
public:
    Options(std::string intro, std::string outro) : m_intro(std::move(intro)), m_outro(std::move(outro)) {}

    void flag(bool *dst, const char *name, const char *alias, const char *help) { m_opts.push_back(Opt(FLAG, name, alias, help, dst)); }
    void text(std::string *dst, const char *name, const char *alias, const char *help, bool required = false)
    {
        Opt o(TEXT, name, alias, help, dst);
        o.required = required;
        m_opts.push_back(o);
    }
    void choice(std::string *dst, const char *name, const char *alias, const char *help, std::vector<std::string> allowed)
    {
        Opt o(TEXT, name, alias, help, dst);
        o.allowedText = std::move(allowed);
        m_opts.push_back(o);
    }
    void number(unsigned int *dst, const char *name, const char *alias, const char *help) { m_opts.push_back(Opt(UINT, name, alias, help, dst)); }
    void number(unsigned int *dst, const char *name, const char *alias, const char *help, unsigned int lo, unsigned int hi)
    {
        Opt o(UINT, name, alias, help, dst);
        o.ranged = true;
        o.lo = lo;
        o.hi = hi;
        m_opts.push_back(o);
    }
    void numberOneOf(unsigned int *dst, const char *name, const char *alias, const char *help, std::vector<unsigned int> allowed)
    {
        Opt o(UINT, name, alias, help, dst);
        o.allowedUint = std::move(allowed);
        m_opts.push_back(o);
    }
    void real(float *dst, const char *name, const char *alias, const char *help) { m_opts.push_back(Opt(REAL, name, alias, help, dst)); }
    void real(float *dst, const char *name, const char *alias, const char *help, float lo, float hi)
    {
        Opt o(REAL, name, alias, help, dst);
        o.ranged = true;
        o.flo = lo;
        o.fhi = hi;
        m_opts.push_back(o);
    }

    void usage() const
    {
        std::fprintf(stderr, "%s\nAvailable options:\n", m_intro.c_str());
        for (const Opt &o : m_opts) {
            const char *ty = o.kind == UINT ? " <int>" : o.kind == REAL ? " <float>" : o.kind == TEXT ? " <string>" : "";
            std::fprintf(stderr, "  %-15s  %-25s\t:  %s\n", (std::string(o.alias) + ty + ",").c_str(), (std::string(o.name) + ty).c_str(), o.help);
        }
        std::fprintf(stderr, "%s\n", m_outro.c_str());
    }

    // false: help was shown, the caller exits with status 1 (as the reference's tools do)
    bool parse(int argc, char **argv)
    {
        std::vector<bool> seen(m_opts.size(), false);
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            if (a == "--help" || a == "-help" || a == "--h" || a == "-h") {
                usage();
                return false;
            }
            size_t k = 0;
            while (k < m_opts.size() && a != m_opts[k].name && a != m_opts[k].alias)
                k++;
            if (k == m_opts.size())
                throw UsageError("The argument '" + a + "' is not a valid input option");
            Opt &o = m_opts[k];
            seen[k] = true;
            if (o.kind == FLAG) {
                *static_cast<bool *>(o.dst) = true;
                continue;
            }
            if (++i >= argc)
                throw UsageError("No value provided for input option '" + a + "'");
            assign(o, a, argv[i]);
        }
        for (size_t k = 0; k < m_opts.size(); k++)
            if (m_opts[k].required && !seen[k])
                throw UsageError(std::string("Missing required option '") + m_opts[k].name + "'");
        return true;
    }

private:
    enum Kind { FLAG, UINT, REAL, TEXT };
    struct Opt {
        Opt(Kind k, const char *n, const char *a, const char *h, void *d) : kind(k), name(n), alias(a), help(h), dst(d) {}
        Kind kind;
        const char *name, *alias, *help;
        void *dst;
        bool required = false, ranged = false;
        unsigned int lo = 0, hi = 0;
        float flo = 0, fhi = 0;
        std::vector<unsigned int> allowedUint;
        std::vector<std::string> allowedText;
    };

    template <typename T, typename V>
    static void requireMember(const std::string &optName, const T &v, const V &allowed)
    {
        if (allowed.empty())
            return;
        for (const auto &x : allowed)
            if (x == v)
                return;
        std::ostringstream m;
        m << "Input '" << v << "' for argument '" << optName << "' is not valid. Valid values are: ";
        for (const auto &x : allowed)
            m << x << " ";
        throw UsageError(m.str());
    }

    static void assign(Opt &o, const std::string &optName, const char *value)
    {
        std::ostringstream m;
        if (o.kind == UINT) {
            const unsigned int v = (unsigned int)std::atoi(value);  // the reference converts with atoi as well
            if (o.ranged && (v < o.lo || v > o.hi)) {
                m << "Argument '" << optName << "' with value '" << v << "' is out of range. Valid range is [" << o.lo << ", " << o.hi << "]";
                throw UsageError(m.str());
            }
            requireMember(optName, v, o.allowedUint);
            *static_cast<unsigned int *>(o.dst) = v;
        } else if (o.kind == REAL) {
            const float v = (float)std::atof(value);
            if (o.ranged && (v < o.flo || v > o.fhi)) {
                m << "Argument '" << optName << "' with value '" << v << "' is out of range. Valid range is [" << o.flo << ", " << o.fhi << "]";
                throw UsageError(m.str());
            }
            *static_cast<float *>(o.dst) = v;
        } else {
            const std::string v = value;
            requireMember(optName, v, o.allowedText);
            *static_cast<std::string *>(o.dst) = v;
        }
    }

    std::string m_intro, m_outro;
    std::vector<Opt> m_opts;
};

inline bool endsWithNoCase(const std::string &s, const std::string &suffix)
{
    if (suffix.size() >= s.size())
        return false;
    return strcasecmp(s.c_str() + s.size() - suffix.size(), suffix.c_str()) == 0;
}

// "<start>:<end>" or "<start>:<step>:<end>" (lumaenc.cpp:77-103): fields that are not given keep their defaults
// The reference's getFrameRange (lumaenc.cpp:79-105 there), behaviour for behaviour: one or two ':' in the whole string, then
// the numbers are read one after the other with strtol, each read starting ONE character behind the end of the previous number
// -- whatever that character is ("1x:5" is therefore rejected: the second read starts at ":5"; "1x5:7" is accepted as 1:5).
inline bool parseFrameRange(const std::string &spec, unsigned int &start, unsigned int &step, unsigned int &end)
{
    size_t nd = 0;
    for (char ch : spec)
        nd += ch == ':';
    if (nd < 1 || nd > 2)
        return false;
    unsigned int *range[3] = {&start, &step, &end};
    const char *p = spec.c_str(), *const last = p + spec.size();
    for (size_t i = 0; i < 3; i += 3 - nd) {
        if (p > last)
            return false;   // (the previous number ended the string: the reference would read past its end here)
        char *stop = nullptr;
        const long v = std::strtol(p, &stop, 10);
        if (stop == p)
            return false;
        *range[i] = (unsigned int)v;
        p = stop + 1;
    }
    return true;
}

}  // namespace lumacli

#endif
