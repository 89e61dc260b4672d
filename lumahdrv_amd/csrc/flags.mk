# Compiler and flags of the device code (its own file so that lumahdrv_amd.capi.kernel_source_sha covers the flags but not
# the rules for tools and examples in the Makefile).
#   -ffp-contract=off : the reference arithmetic is un-contracted fp32 (see luma_device.hpp); fma()
#                       appears only where written explicitly.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
FLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $(EXTRA)
