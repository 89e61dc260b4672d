# Compiler and flags of the device code (its own file so that lumahdrv_amd.capi.kernel_source_sha covers the flags but not
# the rules for tools and examples in the Makefile).
#   -ffp-contract=off : the reference arithmetic is un-contracted fp32 (see luma_device.hpp); fma()
#                       appears only where written explicitly.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
FLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $(EXTRA)
# Per-translation-unit additions (the Makefile appends $(FLAGS_<unit>) to the unit's compile line).
#   lumahip_encode: LLVM's "max-ilp" machine-scheduling strategy.  Same-box A/B of the whole library built either way (round 6,
#   profiles/r06_sched_ab.txt): the HBM-bound encode kernels 1.6 - 1.8 % faster per launch, the decode kernels 0.9 % SLOWER, the
#   VALU-bound YCbCr kernels unchanged -- so it is the encode unit's only.  ("iterative-ilp" crashes this compiler on these units.)
FLAGS_lumahip_encode := -mllvm -amdgpu-sched-strategy=max-ilp
