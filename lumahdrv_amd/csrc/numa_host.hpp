// numa_host.hpp -- where the host side of a context should live on a multi-socket box: the NUMA node of the GPU, that node's
// CPUs, and the two actions taken on them (pin a thread, allocate pinned memory on the node).  Plain C++ (numa_host.cpp).
#pragma once
#include <pthread.h>

#include <cstddef>
#include <string>
#include <vector>

namespace lh {

// "0-63,128-191" -> {0..63, 128..191}; false on anything that is not a cpulist
bool numa_parse_cpulist(const char *s, std::vector<int> &out);
// <sysfs_root>/bus/pci/devices/<bus id>/numa_node (bus id as hipDeviceGetPCIBusId prints it, any case); -1: unknown / not a NUMA box
int numa_node_of_pci(const char *sysfs_root, const char *pci_bus_id);
// <sysfs_root>/devices/system/node/node<N>/cpulist, intersected with `allowed` when that is not empty
bool numa_cpus_of_node(const char *sysfs_root, int node, const std::vector<int> &allowed, std::vector<int> &out);
std::vector<int> numa_allowed_cpus();   // the process's CPUs (affinity mask of the thread-group leader)
bool numa_pin_thread(pthread_t t, const std::vector<int> &cpus);
// the calling thread's memory policy: MPOL_PREFERRED `node`; node < 0: back to the policy the thread had before; false when the kernel refuses
bool numa_prefer_node(int node);

}  // namespace lh
