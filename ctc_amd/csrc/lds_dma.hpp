// Building blocks of the kernels that stage their operands global -> LDS by LDS-DMA
// (global_load_lds_dwordx4): the DMA itself, LDS reads issued from inline assembly, counted waits.
//
// Why the reads are assembly: once a global_load_lds is outstanding the compiler makes every LDS load it
// can see wait for ALL outstanding DMA transfers (it cannot tell that they target another stage), which
// would drain the operand pipeline at every k-step.  The price is that the compiler does not know these
// loads are asynchronous either, so every value read this way passes through lds_wait() (an s_waitcnt
// that names the value as an in/out operand) before its first use.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace ctc {
namespace lds_dma {

// 16 bytes per lane: lane l of the wave lands at lds_wave_base + 16 l, so one instruction fills 1 KB of
// consecutive LDS; WHICH 16 bytes a lane fetches is free (the swizzles of the images are applied there).
__device__ __forceinline__ void dma16(const void *src, unsigned char *lds_wave_base)
{
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// The same with the address split the way the instruction takes it: a wave-uniform 64-bit base in scalar
// registers plus a 32-bit byte offset per lane, so that no vector-ALU instruction forms the address.  Issued
// from assembly (the compiler, given a pointer, adds the two parts on the vector ALU); lds_wave_base is the
// LDS byte address the wave's 1 KB goes to.  The compiler does not count these transfers in its own
// s_waitcnt bookkeeping: the caller waits for them explicitly.  M0 is overwritten (it cannot be named as a
// clobber: the compiler reserves it and only warns), so a kernel that uses this must have no other user of
// M0: no dma16() above (there the compiler sets M0 and may assume it unchanged across this statement), no
// dynamically indexed register array (relative moves go through M0).  tests/test_abi.py disassembles the
// library and checks exactly that for every kernel whose transfers come from here.
__device__ __forceinline__ void dma16_split(const void *uniform_base, unsigned lane_offset, unsigned lds_wave_base)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_offset), "s"(uniform_base), "s"(lds_wave_base)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    typedef __attribute__((address_space(3))) const unsigned char *lptr_t;
    return (unsigned)(uintptr_t)(lptr_t)p;
}
template <int OFF> __device__ __forceinline__ void lds_read64(double &d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_read64(int64_t &d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <typename V> __device__ __forceinline__ void lds_wait(V &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
// the same with up to N younger LDS reads still in flight (LDS reads return in order; the caller knows that
// at least N were issued after the one that fills `a`)
template <int N, typename V> __device__ __forceinline__ void lds_wait_but(V &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <typename V> __device__ __forceinline__ void lds_wait(V &a, V &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lds_wait(double &a0, double &a1, double &a2, double &a3, double &b0, double &b1)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1));
}
__device__ __forceinline__ void lds_wait(double &a0, double &a1, double &a2, double &a3, double &b0, double &b1, double &b2,
                                         double &b3)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

}  // namespace lds_dma
}  // namespace ctc
