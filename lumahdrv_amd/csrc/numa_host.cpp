// numa_host.cpp -- NUMA placement of the host side of a context (numa_host.hpp).  Why: the GPU boxes are two-socket hosts with
// four GPUs per socket; a staging copy that runs on the other socket's cores, or a pinned ring that lives in the other socket's
// memory, crosses the socket interconnect once more per byte -- and with eight shards (lumahip_multi_*) half of them would.
// The batch path this serves replaces the loop at lumaenc.cpp:205-243 of the reference (one thread, one frame at a time).
#include "numa_host.hpp"

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/lumahip.h"

namespace lh {

bool numa_parse_cpulist(const char *s, std::vector<int> &out)
{
    out.clear();
    if (!s)
        return false;
    const char *p = s;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\n' || *p == ',')
            p++;
        if (!*p)
            break;
        if (!isdigit((unsigned char)*p))
            return false;
        char *e = nullptr;
        const long a = strtol(p, &e, 10);
        long b = a;
        p = e;
        if (*p == '-') {
            p++;
            if (!isdigit((unsigned char)*p))
                return false;
            b = strtol(p, &e, 10);
            p = e;
        }
        if (a < 0 || b < a || b > 65535)
            return false;
        for (long c = a; c <= b; c++)
            out.push_back((int)c);
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return !out.empty();
}

static bool read_small_file(const std::string &path, char *buf, size_t cap)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f)
        return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}

int numa_node_of_pci(const char *sysfs_root, const char *pci_bus_id)
{
    if (!sysfs_root || !pci_bus_id || !*pci_bus_id)
        return -1;
    std::string id(pci_bus_id);
    for (auto &ch : id)
        ch = (char)tolower((unsigned char)ch);
    if (id.size() == 7)   // "xx:yy.z": domain 0000
        id = "0000:" + id;
    char buf[64];
    if (!read_small_file(std::string(sysfs_root) + "/bus/pci/devices/" + id + "/numa_node", buf, sizeof buf))
        return -1;
    char *e = nullptr;
    const long v = strtol(buf, &e, 10);
    return (e == buf || v < 0 || v > 4095) ? -1 : (int)v;
}

bool numa_cpus_of_node(const char *sysfs_root, int node, const std::vector<int> &allowed, std::vector<int> &out)
{
    out.clear();
    if (!sysfs_root || node < 0)
        return false;
    char buf[4096];
    if (!read_small_file(std::string(sysfs_root) + "/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof buf))
        return false;
    std::vector<int> all;
    if (!numa_parse_cpulist(buf, all))
        return false;
    if (allowed.empty()) {
        out = all;
    } else {
        for (int c : all)
            if (std::binary_search(allowed.begin(), allowed.end(), c))
                out.push_back(c);
    }
    return !out.empty();
}

// The CPUs the PROCESS was given (the thread-group leader's mask: what `taskset` / a container's cpuset left it), not the
// calling thread's own mask -- a caller that has pinned itself to a core has not thereby confined the library's workers to it.
std::vector<int> numa_allowed_cpus()
{
    std::vector<int> v;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(getpid(), sizeof set, &set) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &set))
                v.push_back(c);
    return v;
}

bool numa_pin_thread(pthread_t t, const std::vector<int> &cpus)
{
    if (cpus.empty())
        return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE)
            CPU_SET(c, &set);
    return pthread_setaffinity_np(t, sizeof set, &set) == 0;
}

// set_mempolicy(2) / get_mempolicy(2) without libnuma: MPOL_DEFAULT = 0, MPOL_PREFERRED = 1; the node mask is a bit field of
// maxnode bits.  The policy the thread had (a caller may run under `numactl --interleave`) is kept and put back.
namespace {
thread_local int t_saved_mode = -1;
thread_local unsigned long t_saved_mask[16];
}  // namespace

bool numa_prefer_node(int node)
{
    if (node < 0) {
        if (t_saved_mode < 0)
            return true;
        const int mode = t_saved_mode;
        t_saved_mode = -1;
        return syscall(SYS_set_mempolicy, mode, mode == 0 ? nullptr : t_saved_mask, mode == 0 ? 0UL : sizeof t_saved_mask * 8) == 0;
    }
    if (node >= 1024)
        return false;
    int mode = 0;
    memset(t_saved_mask, 0, sizeof t_saved_mask);
    if (syscall(SYS_get_mempolicy, &mode, t_saved_mask, sizeof t_saved_mask * 8, nullptr, 0UL) != 0)
        return false;   // cannot tell what to restore: leave the policy alone
    unsigned long mask[16];
    memset(mask, 0, sizeof mask);
    mask[node / (8 * sizeof(unsigned long))] |= 1UL << (node % (8 * sizeof(unsigned long)));
    if (syscall(SYS_set_mempolicy, 1, mask, sizeof mask * 8) != 0)
        return false;
    t_saved_mode = mode;   // (with its mode flags, as get_mempolicy reports them)
    return true;
}

}  // namespace lh

// Host-only (no GPU, no context): the placement the library derives for a GPU from a sysfs tree -- its NUMA node and the CPUs
// its staging threads are pinned to.  `sysfs_root` NULL = "/sys"; the tests hand in a fabricated tree.
extern "C" int lumahip_numa_plan_host(const char *sysfs_root, const char *pci_bus_id, const char *allowed_cpulist, int *node, int *cpus,
                                      int cap, int *ncpus)
{
    if (!pci_bus_id || !node || !ncpus || cap < 0 || (cap > 0 && !cpus))
        return LUMAHIP_ERR_ARG;
    const char *root = sysfs_root ? sysfs_root : "/sys";
    *node = lh::numa_node_of_pci(root, pci_bus_id);
    *ncpus = 0;
    if (*node < 0)
        return LUMAHIP_OK;   // not a NUMA box (or an unknown device): nothing to plan
    std::vector<int> allowed, out;
    if (allowed_cpulist && *allowed_cpulist && !lh::numa_parse_cpulist(allowed_cpulist, allowed))
        return LUMAHIP_ERR_ARG;
    if (!lh::numa_cpus_of_node(root, *node, allowed, out))
        return LUMAHIP_OK;
    *ncpus = (int)out.size();
    for (int i = 0; i < cap && i < (int)out.size(); i++)
        cpus[i] = out[i];
    return LUMAHIP_OK;
}
