// LDS-DMA (global_load_lds_dwordx4) issued through inline asm.
//
// 16 bytes per lane from global address `g` to LDS at the wave-uniform byte address `ldsAddr` + lane*16.  With the
// builtin (__builtin_amdgcn_global_load_lds) the compiler tracks an in-flight LDS write that may alias every later
// ds_read of the same wave and then waits lgkmcnt(0) - including the reads it has just issued - in front of every
// use of LDS data; issued as asm, the ds_read waits are counted (lgkmcnt(N)).  The kernel must wait vmcnt(0) itself
// before the barrier that publishes the staged data (the compiler does not know about these loads).
// m0 is a reserved register and cannot be named as a clobber; nothing else in these kernels uses it.
#pragma once
#include <hip/hip_runtime.h>

namespace unet {

__device__ __forceinline__ void lds_dma16(const void* g, unsigned ldsAddr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(ldsAddr), "v"(g) : "memory");
}

// LDS byte address of a __shared__ object
template <class T>
__device__ __forceinline__ unsigned lds_address(T* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) void*)p;
}

}  // namespace unet
