// lds_dma.h — direct-to-LDS buffer loads and the counted-wait / raw-barrier helpers shared by the conv kernels.
#pragma once
#include "common.h"

namespace mi355 {
namespace {

#define MI355_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define MI355_LDS_BARRIER()                                \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)

typedef __attribute__((address_space(3))) void* lptr_t;

typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor over [p, p+bytes): stride 0, 32-bit byte offsets, out-of-range reads return 0
__device__ __forceinline__ i32x4 make_srd(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// One 1 KiB LDS-DMA piece: lane l's 16 bytes at byte offset `voff` of the buffer land at LDS address lds_dst + 16*l
// (lds_dst wave-uniform); voff >= num_records => zeros (verified on MI355X by tools/micro/blds_test.hip).
// Inline asm on purpose: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first VALU write of the address
// registers of a __builtin_amdgcn_*_load_lds, which serialises the very loads this pipeline keeps in flight; the
// hardware reads the offset at issue, so no wait is needed.
__device__ __forceinline__ void blds16(i32x4 srd, unsigned voff, unsigned lds_dst) {
  // M0 (the LDS base of the DMA) is declared clobbered instead of saved/restored: two scalar instructions less per piece
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(lds_dst)
               : "memory", "m0");
}
// The same with the source offset split into per-lane `voff` + wave-uniform `soff` (the instruction's scalar offset: no
// VALU add) and the LDS destination formed as lds_base + IMM by the s_add that writes M0.
template <int IMM>
__device__ __forceinline__ void blds16o(i32x4 srd, unsigned voff, unsigned soff, unsigned lds_base) {
  asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_base), "i"(IMM)
               : "memory", "m0", "scc");
}
template <int IMM>
__device__ __forceinline__ void blds16z(i32x4 srd, unsigned voff, unsigned lds_base) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(lds_base), "i"(IMM)
               : "memory", "m0", "scc");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(lptr_t)p;
}

}  // namespace
}  // namespace mi355
