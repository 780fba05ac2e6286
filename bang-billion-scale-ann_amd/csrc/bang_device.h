// bang_device.h -- device-side helpers shared by the kernel translation units (bang_kernels.hip: per-iteration kernels and the
// round-1 persistent kernel; bang_search.hip: the query-resident search kernel).  Not public.
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.
#ifndef BANG_DEVICE_H_
#define BANG_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bang_c.h"
#include "bang_internal.h"

#define WAVE 64
// A pointer the kernel rebuilds from an integer (IterArgs, a slice-table entry, a kernarg field read with KARG) is a GENERIC pointer to the compiler and every
// access through it a flat_ instruction -- which counts in lgkmcnt as well as vmcnt, so that every wait for an LDS result also waits for the loads in flight.
// Typed as global (address space 1) it is a global_ instruction again (and may take its base from scalar registers).
#define GAS __attribute__((address_space(1)))
#define BIG_DIST ((float)3.402823E+38)  // bang_search.cu:1406,1484

typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef const float __attribute__((address_space(4))) * cfloat_p;  // constant AS: scalar loads

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash1(uint32_t x) {  // hashFn1_d :1168-1178
  uint64_t h = 0xcbf29ce4ull;
  h = (h ^ (uint64_t)(x & 0xff)) * 0x01000193ull;
  h = (h ^ (uint64_t)((x >> 8) & 0xff)) * 0x01000193ull;
  h = (h ^ (uint64_t)((x >> 16) & 0xff)) * 0x01000193ull;
  h = (h ^ (uint64_t)((x >> 24) & 0xff)) * 0x01000193ull;
  return (uint32_t)(h % BANG_BF_ENTRIES);
}
__device__ __forceinline__ uint32_t hash2(uint32_t x) {  // hashFn2_d :1180-1189
  uint64_t h = 0x84222325ull;
  h = (h ^ (uint64_t)(x & 0xff)) * 0x1B3ull;
  h = (h ^ (uint64_t)((x >> 8) & 0xff)) * 0x1B3ull;
  h = (h ^ (uint64_t)((x >> 16) & 0xff)) * 0x1B3ull;
  h = (h ^ (uint64_t)((x >> 24) & 0xff)) * 0x1B3ull;
  return (uint32_t)(h % BANG_BF_ENTRIES);
}

__device__ __forceinline__ void bloom_set(uint32_t* w, uint32_t bit) {
  (void)__hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void wave_sync() {
  // Intra-wave LDS hand-off: a wave executes its LDS instructions in program order, so a plain
  // write followed by a read needs no fence; this only stops the compiler from moving code across.
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {  // popcount of mask bits below my lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// ------------------------------------------------------------------------------------------
// K2 core: distance of one neighbour (one lane) -- pivot-stationary form
// ------------------------------------------------------------------------------------------
// Canonical float order (bang_search.cu:1229-1239): eight partial sums
// s_l = (((0 + t_l) + t_{l+8}) + ...), then ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)).
// t_c = LUT[c][code_c] is recomputed as the fmaf chain of populate_pqDist_par (:1118-1128);
// zero padding of a chunk to PSZ dims adds fmaf(0,0,t) == t, i.e. nothing.
// NHI > 0 (PSZ == 2 only): exact-size table -- chunks [0, NHI) hold 2 floats per entry, the rest 1.  A 1-dim chunk skips the
// second term, which is what the padded table computes for it (fmaf(0, 0, t) == t).  NHI is a template parameter so that every
// entry address stays "code * size + immediate" (a runtime split costs an address register per chunk: measured as 100-200
// spilled VGPRs in every 2-float instance).
// QC: where the centred query comes from -- anything indexable by the padded dimension number.  `cfloat_p` (constant address
// space: scalar loads) or QcRegs (the wave's registers, one v_readlane per element: no memory wait at all; scalar loads return
// out of order, so every use of one forces `s_waitcnt lgkmcnt(0)` and with it a drain of the LDS reads in flight).
template <int NV>
struct QcRegs {
  float v[NV];                                  // lane l of v[r] holds qc[64 r + l]
  __device__ __forceinline__ float operator[](uint32_t i) const {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v[i >> 6]), (int)(i & 63u)));
  }
};

// QcRow16: the centred query replicated per 16-lane row -- lane l of v[r] holds qc[16 r + (l & 15)] -- so that "pivot - query" is ONE
// instruction, v_subrev_f32 with a DPP row broadcast of the query operand, instead of v_readlane + v_sub: two of the nine VALU instructions
// a 2-dim chunk costs (bfe, shift, 2 readlane, 2 sub, mul, fma, add).  ceil(QW / 16) registers per query instead of ceil(QW / 64).
// A DPP operand is fetched from ANOTHER lane: every lane of the wave must be executing (an inactive source lane yields 0).
template <int NR>
struct QcRow16 {
  float v[NR];
};
template <int N>
__device__ __forceinline__ float sub_row_bcast(float p, float qreg) {     // p - (lane N of qreg's 16-lane row)
  float d;
  asm("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(qreg), "v"(p), "n"(N));
  return d;
}
template <class QC>
__device__ __forceinline__ float qc_sub(const QC& qc, float p, uint32_t i) { return p - qc[i]; }
template <int NR>
__device__ __forceinline__ float qc_sub(const QcRow16<NR>& qc, float p, uint32_t i) {
  const float r = qc.v[i >> 4];
  switch (i & 15u) {                                   // i is a constant after unrolling: one case survives
    case 0: return sub_row_bcast<0>(p, r);   case 1: return sub_row_bcast<1>(p, r);   case 2: return sub_row_bcast<2>(p, r);
    case 3: return sub_row_bcast<3>(p, r);   case 4: return sub_row_bcast<4>(p, r);   case 5: return sub_row_bcast<5>(p, r);
    case 6: return sub_row_bcast<6>(p, r);   case 7: return sub_row_bcast<7>(p, r);   case 8: return sub_row_bcast<8>(p, r);
    case 9: return sub_row_bcast<9>(p, r);   case 10: return sub_row_bcast<10>(p, r); case 11: return sub_row_bcast<11>(p, r);
    case 12: return sub_row_bcast<12>(p, r); case 13: return sub_row_bcast<13>(p, r); case 14: return sub_row_bcast<14>(p, r);
    default: return sub_row_bcast<15>(p, r);
  }
}
template <int PSZ, int NHI, class QC>
__device__ __forceinline__ float lut_entry(const float* __restrict__ piv_lds, const QC& qc, uint32_t c, uint32_t code) {
  float t = 0.0f;
  if (PSZ == 2 && NHI > 0) {
    if (c < (uint32_t)NHI) {
      const float2 p = *(const float2*)(piv_lds + c * 512u + code * 2u);
      const float d0 = qc_sub(qc, p.x, c * 2 + 0);
      t = __builtin_fmaf(d0, d0, t);
      const float d1 = qc_sub(qc, p.y, c * 2 + 1);
      t = __builtin_fmaf(d1, d1, t);
    } else {
      const float d0 = qc_sub(qc, piv_lds[(uint32_t)NHI * 256u + c * 256u + code], c * 2 + 0);
      t = __builtin_fmaf(d0, d0, t);
    }
    return t;
  }
  const float* e = piv_lds + ((size_t)c * 256 + code) * PSZ;
  if (PSZ == 1) {
    const float d = qc_sub(qc, e[0], c);
    t = __builtin_fmaf(d, d, t);
  } else if (PSZ == 2) {
    const float2 p = *(const float2*)e;
    const float d0 = qc_sub(qc, p.x, c * 2 + 0);
    t = __builtin_fmaf(d0, d0, t);
    const float d1 = qc_sub(qc, p.y, c * 2 + 1);
    t = __builtin_fmaf(d1, d1, t);
  } else {
#pragma unroll
    for (int i = 0; i < PSZ; i += 4) {
      const float4 p = *(const float4*)(e + i);
      const float d0 = qc_sub(qc, p.x, c * PSZ + i + 0);
      t = __builtin_fmaf(d0, d0, t);
      const float d1 = qc_sub(qc, p.y, c * PSZ + i + 1);
      t = __builtin_fmaf(d1, d1, t);
      const float d2 = qc_sub(qc, p.z, c * PSZ + i + 2);
      t = __builtin_fmaf(d2, d2, t);
      const float d3 = qc_sub(qc, p.w, c * PSZ + i + 3);
      t = __builtin_fmaf(d3, d3, t);
    }
  }
  return t;
}

// NDW = number of code dwords per row after padding the chunk count to MP = 4*NDW (the padding
// chunks are all-zero in the packed pivot table and in qc, so they add fmaf(0,0,0) = +0).
// ALIGNED = (m % 4 == 0): rows start dword-aligned, no funnel shift needed.
// A code row is fetched with 16-byte loads from its 4-byte-aligned base (pq_row_load) and consumed by
// straight-line code (pq_row_reduce) so that the compiler can batch the LDS reads and scalar loads;
// splitting load from use lets the caller put another query's row in flight first.
// reinterpret a byte pointer of either address space as a pointer to T of the same address space
template <class T> __device__ __forceinline__ T* ptr_as(const uint8_t* q) { return (T*)q; }
template <class T> __device__ __forceinline__ T GAS* ptr_as(const uint8_t GAS* q) { return (T GAS*)q; }
template <int NDW, bool ALIGNED>
struct PqRow {
  static constexpr int NLOAD = ALIGNED ? NDW : NDW + 1;   // dwords needed from the aligned base
  static constexpr int NX4 = (NLOAD + 3) / 4;
  uint32_t w[NX4 * 4 + 1];
  uint32_t sh;
};

// (CP: `const uint8_t*` or `const uint8_t GAS*`)
template <int NDW, bool ALIGNED, class CP>
__device__ __forceinline__ void pq_row_load(PqRow<NDW, ALIGNED>& r, CP codes, uint32_t m, uint32_t id) {
  const uint64_t a = (uint64_t)id * m;  // 64-bit row offset, :1232
  r.sh = (uint32_t)a & 3u;
  const auto p = ptr_as<const u32x4a>(codes + (a & ~3ull));
#pragma unroll
  for (int i = 0; i < PqRow<NDW, ALIGNED>::NX4; ++i) {
    const u32x4a v = p[i];
    r.w[4 * i + 0] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
  }
  r.w[PqRow<NDW, ALIGNED>::NX4 * 4] = 0;
}

// COOPERATIVE fetch of the code rows of a wave's <= 64 survivors.  With per-lane loads (pq_row_load) every one of the NX4 load
// instructions of a wave touches 64 different lines -- 5 x 64 line look-ups for 64 rows of 70 bytes, each row's two or three lines
// looked up again by every instruction -- and the memory pipeline retires ~85 G such look-ups a second: 17 G rows/s
// (tools/dev/row_fetch_bench.hip).  Here P = NX4 ADJACENT lanes ask for the P 16-byte pieces of ONE row in one instruction, so the
// pieces of a row merge into one request per line: 30 G rows/s for packed 70-byte rows, 47 G rows/s where a row never leaves its
// 128-byte line (code_stride = 128) -- the random-access rate of the memory system itself.  A wave instruction covers RPI = 64 / P
// rows; NI of them cover 64.  The pieces then change hands through 1 KB of the wave's LDS (RPI rows at a time: the lanes that loaded
// a row store their pieces, the lane that evaluates the row reads all P back), so the reduce below sees the same PqRow as before.
template <int NDW, bool ALIGNED>
struct CoopFetch {
  typedef PqRow<NDW, ALIGNED> Row;
  static constexpr int P = Row::NX4;                   // 16-byte pieces per row
  static constexpr int RPI = 64 / P;                   // rows per wave instruction
  static constexpr int NI = (64 + RPI - 1) / RPI;      // wave instructions for 64 rows
  static constexpr int LDS_WORDS = RPI * P * 4;        // staging: one instruction's worth of pieces (<= 256 words)
  u32x4a v[NI];
  // id: the row this lane will evaluate (valid for lane < n); every lane of the wave must be executing
  template <class CP>                                  // `const uint8_t*` or `const uint8_t GAS*`
  __device__ __forceinline__ void issue(CP codes, uint32_t stride, uint32_t id, uint32_t n, int lane) {
    const uint32_t slot = (uint32_t)lane / (uint32_t)P, piece = (uint32_t)lane % (uint32_t)P;
    // all the ids first (NI independent ds_bpermute in flight), then the loads: interleaved, every load waits for its own id's round
    // trip through the LDS crossbar
    uint32_t rid[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) rid[j] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)j * RPI + slot) & 63u) << 2), (int)id);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint32_t rr = (uint32_t)j * RPI + slot;                                       // the row (= lane number of its evaluator)
      const bool ok = slot < (uint32_t)RPI && rr < n;
      // a lane with no row to fetch reads row 0 instead of sitting the instruction out: a load under a branch makes the compiler's
      // count of outstanding memory operations inexact, and every later wait for an OLDER load then becomes a wait for everything
      const uint64_t a = ok ? (uint64_t)rid[j] * stride : 0ull;
      const auto p = ptr_as<const u32x4a>(codes + (a & ~3ull)) + (ok ? piece : 0u);
      // non-temporal: a code row is read once and never again -- marked so, it does not push the resident queries' filter lines out of L2 / the
      // Infinity Cache as fast (10 K SIFT1B-shape batch 8.15 -> 7.98 ms).  Round 3 had measured the hint slower with per-lane loads: an `nt` load
      // by-passes L1, and a lane walking its own row then turned every one of its five loads into an L2 request; here the P lanes of a row ask
      // for its pieces in ONE instruction and the address coalescer makes that one request per line with or without L1.
      v[j] = __builtin_nontemporal_load(p);
    }
  }
  // BUF_WORDS: words of staging the wave has (>= LDS_WORDS: one wave instruction's pieces change hands at once; less: in ROUNDS rounds of RR rows --
  // the same stores and reads per lane, only more hand-overs; what lets 16 waves share the LDS the pivot table leaves)
  template <int BUF_WORDS = LDS_WORDS>
  __device__ __forceinline__ void collect(Row& r, uint32_t* buf /* BUF_WORDS words of the wave's LDS, 16-byte aligned */, uint32_t stride,
                                          uint32_t id, int lane) {
    constexpr int RR = (BUF_WORDS >= LDS_WORDS) ? RPI : BUF_WORDS / (P * 4);          // rows per round
    constexpr int ROUNDS = (RPI + RR - 1) / RR;
    static_assert(RR >= 1, "the staging area holds at least one row");
    r.sh = ((uint32_t)id * stride) & 3u;                // (low two bits of the 64-bit row offset)
    const uint32_t slot = (uint32_t)lane / (uint32_t)P;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        const uint32_t s0 = (uint32_t)rd * RR;                                           // first slot of this round
        if (slot >= s0 && slot < s0 + (uint32_t)RR && slot < (uint32_t)RPI) *(u32x4a*)(buf + 4 * ((uint32_t)lane - s0 * P)) = v[j];   // (lane = slot * P + piece: contiguous)
        wave_sync();
        const uint32_t k = (uint32_t)lane - (uint32_t)j * RPI - s0;                      // my row within this round's batch
        if (k < (uint32_t)RR && k + s0 < (uint32_t)RPI) {
#pragma unroll
          for (int q = 0; q < P; ++q) {
            const u32x4a t = *(const u32x4a*)(buf + (k * P + q) * 4);
            r.w[4 * q + 0] = t.x; r.w[4 * q + 1] = t.y; r.w[4 * q + 2] = t.z; r.w[4 * q + 3] = t.w;
          }
        }
        wave_sync();
      }
    }
    r.w[Row::NX4 * 4] = 0;
    // every piece register is read once more by ALL lanes: the stores above sit behind `slot < RPI`, so on the other path the
    // compiler still counts these loads as outstanding and would wait for "them" -- and for every store issued since -- wherever
    // one of the registers is reused
#pragma unroll
    for (int j = 0; j < NI; ++j) asm volatile("" : : "v"(v[j]));
  }
};

// SB > 0: the row is consumed in segments of SB code dwords; a dependency fence between segments keeps the compiler from hoisting
// the LDS reads of the whole row (72 chunks of the 70/74-chunk layouts) in front of the first add -- that is what pushed those
// instances past 128 VGPRs.
template <int PSZ, int NDW, bool ALIGNED, int NHI, int SB = 0, class QC = cfloat_p>
__device__ __forceinline__ float pq_row_reduce(const PqRow<NDW, ALIGNED>& r, const float* __restrict__ piv_lds,
                                               const QC& qc) {
  float s[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) s[l] = 0.0f;
#pragma unroll
  for (int k = 0; k < NDW; ++k) {
    if constexpr (SB > 0) {
      if (k > 0 && (k % SB) == 0) {
        // Every partial sum and every code dword still to be consumed passes through an empty asm: nothing of the next segment
        // (byte extraction -> LDS address -> read) can be scheduled in front of the last add of this one.
        asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
        uint32_t* w = const_cast<uint32_t*>(r.w);
#pragma unroll
        for (int kk = k; kk < NDW + (ALIGNED ? 0 : 1) && kk < k + SB + 1; ++kk) asm volatile("" : "+v"(w[kk]));
      }
    }
    const uint32_t dw = ALIGNED ? r.w[k] : __builtin_amdgcn_alignbyte(r.w[k + 1], r.w[k], r.sh);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t c = 4 * k + b;
      const uint32_t code = (dw >> (8 * b)) & 0xffu;
      const float t = lut_entry<PSZ, NHI>(piv_lds, qc, c, code);
      s[c & 7] = s[c & 7] + t;
    }
  }
  const float x = (s[0] + s[1]) + (s[2] + s[3]);
  const float y = (s[4] + s[5]) + (s[6] + s[7]);
  return x + y;
}


// The same reduce as a two-stage software pipeline over groups of 8 chunks: the pivot entries of group g+1 are requested from LDS
// before the arithmetic of group g starts.  Left to itself the scheduler keeps four LDS reads in flight and waits for the first right
// behind the fourth (it schedules for register pressure): 18 exposed LDS round trips per 72-chunk row, ~1.7 us per iteration of a
// wave that has its SIMD to itself.  A group is 8 chunks so that every partial sum s[l] takes exactly one term per group: the
// canonical order (s_l = ((t_l + t_{l+8}) + t_{l+16}) ...) is the plain one (groups of 4 or 2 chunks for the wider entries: the
// partial sums still take their terms in chunk order).
template <int PSZ, int NHI>
struct PivEntry {                                       // one LUT entry's pivot floats (PSZ = 1, 2 or 4 per read; PSZ = 8: two halves)
  float v[PSZ];
};
template <int PSZ, int NHI>
__device__ __forceinline__ void piv_entry_load(PivEntry<PSZ, NHI>& e, const float* __restrict__ piv_lds, uint32_t c, uint32_t code) {
  if (PSZ == 2 && NHI > 0) {
    if (c < (uint32_t)NHI) { const float2 p = *(const float2*)(piv_lds + c * 512u + code * 2u); e.v[0] = p.x; e.v[1] = p.y; }
    else { e.v[0] = piv_lds[(uint32_t)NHI * 256u + c * 256u + code]; e.v[1] = 0.0f; }
    return;
  }
  const float* a = piv_lds + ((size_t)c * 256 + code) * PSZ;
  if (PSZ == 1) e.v[0] = a[0];
  else if (PSZ == 2) { const float2 p = *(const float2*)a; e.v[0] = p.x; e.v[1] = p.y; }
  else {
#pragma unroll
    for (int i = 0; i < PSZ; i += 4) { const float4 p = *(const float4*)(a + i); e.v[i] = p.x; e.v[i + 1] = p.y; e.v[i + 2] = p.z; e.v[i + 3] = p.w; }
  }
}
template <int PSZ, int NHI, class QC>
__device__ __forceinline__ float piv_entry_eval(const PivEntry<PSZ, NHI>& e, const QC& qc, uint32_t c) {     // == lut_entry()
  float t = 0.0f;
  constexpr int ND = PSZ;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    if (PSZ == 2 && NHI > 0 && i == 1 && c >= (uint32_t)NHI) break;                   // 1-dim chunk of the exact-size table
    const float d = qc_sub(qc, e.v[i], c * PSZ + i);
    t = __builtin_fmaf(d, d, t);
  }
  return t;
}
template <int N>
__device__ __forceinline__ void reg_fence(float* v) {   // an empty asm every value passes through: what is computed from them comes after it
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}
template <int PSZ, int NDW, bool ALIGNED, int NHI, class QC = cfloat_p>
__device__ __forceinline__ float pq_row_reduce_pipe(const PqRow<NDW, ALIGNED>& r, const float* __restrict__ piv_lds, const QC& qc) {
  constexpr int G = PSZ <= 2 ? 8 : PSZ == 4 ? 4 : 2;    // chunks per group: <= 16 pivot floats per lane and group
  constexpr int NC = 4 * NDW;
  constexpr int NG = (NC + G - 1) / G;
  float s[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) s[l] = 0.0f;
  PivEntry<PSZ, NHI> e[2][G];
  uint32_t* w = const_cast<uint32_t*>(r.w);
  auto load_group = [&](int g, PivEntry<PSZ, NHI>* dst) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = g * G + j;
      if (c < NC) {
        const int k = c >> 2;
        const uint32_t dw = ALIGNED ? w[k] : __builtin_amdgcn_alignbyte(w[k + 1], w[k], r.sh);
        piv_entry_load<PSZ, NHI>(dst[j], piv_lds, (uint32_t)c, (dw >> (8 * (c & 3))) & 0xffu);
      }
    }
  };
  // the fence in front of group g's arithmetic: the partial sums, the entries of group g (their reads have returned) and the code
  // dwords group g+1 is addressed with all pass through it -- the reads of group g+1 cannot be issued earlier (two groups of entries
  // live at a time, not the whole row's) and nothing of group g is computed before its entries are there
  auto fence = [&](int g) {
    reg_fence<8>(s);
    if (g < NG) {
#pragma unroll
      for (int j = 0; j < G; ++j) reg_fence<PSZ>(e[g & 1][j].v);
    }
#pragma unroll
    for (int c = (g + 1) * G; c < (g + 2) * G && c < NC; c += 4) {
      asm volatile("" : "+v"(w[c >> 2]));
      if (!ALIGNED) asm volatile("" : "+v"(w[(c >> 2) + 1]));
    }
  };
  load_group(0, e[0]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    fence(g);
    if (g + 1 < NG) load_group(g + 1, e[(g + 1) & 1]);
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = g * G + j;
      if (c < NC) s[c & 7] = s[c & 7] + piv_entry_eval<PSZ, NHI>(e[g & 1][j], qc, (uint32_t)c);
    }
  }
  const float x = (s[0] + s[1]) + (s[2] + s[3]);
  const float y = (s[4] + s[5]) + (s[6] + s[7]);
  return x + y;
}

__device__ __forceinline__ uint32_t lower_bound_lds(const float* arr, uint32_t hi, float target) {  // :1718-1732
  uint32_t lo = 0;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (target <= arr[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}
__device__ __forceinline__ uint32_t upper_bound_lds(const float* arr, uint32_t hi, float target) {  // :1735-1749
  uint32_t lo = 0;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (target >= arr[mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------------------------------------
// wave-level helpers of the search kernel
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_bypass_l1(const uint32_t* p) {   // global_load_dword sc1: served by L2, never by a stale L1 line
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_bypass_l1(const uint32_t GAS* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wave-wide minimum of a 64-bit key {hi, lo} (lexicographic, unsigned), returned to every lane.  Six DPP steps (quad swaps, row
// half-mirror / mirror, two row broadcasts) instead of six ds_bpermute round trips through the LDS crossbar per operand.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_min_step(uint32_t& hi, uint32_t& lo) {
  const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
  const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
  const bool take = ohi < hi || (ohi == hi && olo < lo);
  if (take) { hi = ohi; lo = olo; }
}
__device__ __forceinline__ void wave_min_key(uint32_t& hi, uint32_t& lo) {
  dpp_min_step<0xB1, 0xf>(hi, lo);      // quad_perm [1,0,3,2]
  dpp_min_step<0x4E, 0xf>(hi, lo);      // quad_perm [2,3,0,1]
  dpp_min_step<0x141, 0xf>(hi, lo);     // row_half_mirror
  dpp_min_step<0x140, 0xf>(hi, lo);     // row_mirror            -> every lane of a 16-lane row holds the row's minimum
  dpp_min_step<0x142, 0xa>(hi, lo);     // row_bcast15 into rows 1 and 3
  dpp_min_step<0x143, 0xc>(hi, lo);     // row_bcast31 into rows 2 and 3 -> lane 63 holds the minimum of the wave
  hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
  lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
}


// ---------------------------------------------------------------------------------------------------------------------
// K6 + K7 by the wave that finished the query (compute_L2Dist :1254-1299, compute_NearestNeighbours :1312-1368), 8-bit vectors
// ---------------------------------------------------------------------------------------------------------------------
// The candidate log of the query (n <= L + 50 expanded nodes, written by this wave) is read back, G = D / 16 adjacent lanes fetch one
// candidate's vector in one instruction (16 bytes each: one request per line), eight such instructions in flight.  The squared distance
// is an INTEGER below 2^24 (256 x 255^2), so sum(a - b)^2 = sum a^2 - 2 sum a b + sum b^2 by v_dot4 is exact, and its float image is the
// value the reference's ascending fmaf chain over float(a - b) produces (every partial sum of that chain is such an integer too:
// rerank_kernel, bang_kernels.hip).  Distances go to LDS (the query's worklist and scratch are dead); the k smallest {distance bits, index}
// keys -- ties keep expansion order (:1330-1363) -- land in ids_out [Q][k] / dists_out [rank][Q], a short log is padded with UINT64_MAX /
// BIG_DIST (CANON 8).
template <bool SIGNED>
__device__ __forceinline__ int dot4_8(uint32_t a, uint32_t b, int c) {
  if (SIGNED) return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
  return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)c, false);
}
struct RerankArgs8 {                                    // bang_search_params.rr_*, as the kernel read them at the query's end (global address space: GAS)
  const uint8_t GAS* queries; const uint8_t GAS* vec_base; uint64_t vec_stride; uint64_t GAS* ids_out; float GAS* dists_out; const uint32_t GAS* cand;
  uint32_t D, k, q0, Q_total;
};
// K7 of the fused re-rank (compute_NearestNeighbours :1312-1368): e[0, n) = the exact distances' bit patterns in LDS, in expansion order
__device__ __forceinline__ void wave_topk(const RerankArgs8& p, size_t qabs, uint32_t n, uint32_t* e, int lane) {
  const uint32_t GAS* cand = p.cand;
  // K7: the k smallest {distance bits, index} keys, in order, by repeated wave-wide arg-min -- k rounds of six DPP steps instead of n^2 / 64
  // compares per lane.  Lane l keeps the minimum over ITS candidates (index = l mod 64); the round's winner is struck out by its owner, which
  // re-reads its (<= 9) candidates.  Result r of a chunk of 64 waits in lane r until the chunk is written out.
  uint32_t my_hi = 0xFFFFFFFFu, my_lo = 0xFFFFFFFFu;
  auto own_min = [&]() {
    my_hi = 0xFFFFFFFFu; my_lo = 0xFFFFFFFFu;
    for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) {          // ascending index, strict '<': ties keep expansion order (:1330-1363)
      const uint32_t d = e[i];
      if (d < my_hi) { my_hi = d; my_lo = i; }
    }
  };
  own_min();
  const uint32_t kk = p.k < n ? p.k : n;
  for (uint32_t r0 = 0; r0 < kk; r0 += WAVE) {                     // (uniform)
    uint32_t res_i = 0, res_d = 0;
    const uint32_t rn = kk - r0 < WAVE ? kk - r0 : WAVE;
    for (uint32_t r = 0; r < rn; ++r) {
      uint32_t hi = my_hi, lo = my_lo;
      wave_min_key(hi, lo);                                          // (every lane gets the winner)
      if ((uint32_t)lane == r) { res_i = lo; res_d = hi; }
      if ((lo & 63u) == (uint32_t)lane) { e[lo] = 0xFFFFFFFFu; own_min(); }
    }
    if ((uint32_t)lane < rn) {
      p.ids_out[qabs * p.k + r0 + lane] = (uint64_t)ld_bypass_l1(cand + res_i);               // [Q][k] u64 :1366
      p.dists_out[(size_t)(r0 + (uint32_t)lane) * p.Q_total + qabs] = __uint_as_float(res_d);  // [rank][Q] :999,1297
    }
  }
  for (uint32_t r = n + (uint32_t)lane; r < p.k; r += WAVE) {                      // CANON 8
    p.ids_out[qabs * p.k + r] = ~0ull;
    p.dists_out[(size_t)r * p.Q_total + qabs] = BIG_DIST;
  }
  wave_sync();
}

template <bool SIGNED>
__device__ __forceinline__ void wave_rerank8(const RerankArgs8& p, uint32_t q, uint32_t n, uint32_t* e /* LDS, n words */, int lane) {
  constexpr int U = 4;                                            // vector fetches in flight per lane
  const uint32_t D = p.D, G = D >> 4, per = 64u / G;              // lanes per candidate, candidates per wave instruction
  const uint32_t sub = (uint32_t)lane & (G - 1u), slot = (uint32_t)lane / G;
  const uint32_t GAS* cand = p.cand;
  const size_t qabs = (size_t)p.q0 + q;
  const u32x4a qw = *(const u32x4a GAS*)(p.queries + qabs * D + 16u * sub);
  const int qq = dot4_8<SIGNED>(qw.x, qw.x, dot4_8<SIGNED>(qw.y, qw.y, dot4_8<SIGNED>(qw.z, qw.z, dot4_8<SIGNED>(qw.w, qw.w, 0))));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the log's last word has reached L2 (it is read back past L1)
  for (uint32_t i0 = 0; i0 < n; i0 += per * U) {                  // (uniform)
    uint32_t id[U];
    u32x4a v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + (uint32_t)u * per + slot;
      id[u] = ld_bypass_l1(cand + (i < n ? i : 0u));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *(const u32x4a GAS*)(p.vec_base + (uint64_t)id[u] * p.vec_stride + 16u * sub);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + (uint32_t)u * per + slot;
      int vv = dot4_8<SIGNED>(v[u].x, v[u].x, dot4_8<SIGNED>(v[u].y, v[u].y, dot4_8<SIGNED>(v[u].z, v[u].z, dot4_8<SIGNED>(v[u].w, v[u].w, qq))));
      const int vq = dot4_8<SIGNED>(v[u].x, qw.x, dot4_8<SIGNED>(v[u].y, qw.y, dot4_8<SIGNED>(v[u].z, qw.z, dot4_8<SIGNED>(v[u].w, qw.w, 0))));
      vv -= 2 * vq;
      for (uint32_t off = 1; off < G; off <<= 1) vv += __shfl_xor(vv, (int)off);       // (the G lanes of a candidate are adjacent)
      if (i < n && sub == 0u) e[i] = __float_as_uint((float)vv);
    }
  }
  wave_sync();
  wave_topk(p, qabs, n, e, lane);
}

// The same for FLOAT vectors (compute_L2Dist<float> :1254-1299): the ascending fmaf chain over the D dimensions is one serial chain per candidate
// (float addition does not re-associate), so a lane runs the chain of ITS candidate -- 64 candidates per round -- over 16-byte loads of its
// candidate's vector, RF in flight (a vector's lines are looked up once per 16 bytes: the L1 serves all but the first of a line; the standalone
// rerank_kernel hands a vector over through LDS instead, which a wave at the end of its query has no room for).  The query sits in <= 4
// registers of the wave (lane l of qr[t] = element 64 t + l), element j is read with v_readlane.  Same bits as rerank_kernel<float>.
template <int RF>                                        // 16-byte vector loads in flight per lane
__device__ __forceinline__ void wave_rerank_f32(const RerankArgs8& p, uint32_t q, uint32_t n, uint32_t* e /* LDS, n words */, int lane) {
  const uint32_t D = p.D;                                         // a multiple of 4, <= 256
  const uint32_t GAS* cand = p.cand;
  const size_t qabs = (size_t)p.q0 + q;
  const float GAS* qsrc = (const float GAS*)p.queries + qabs * D;
  float qr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { const uint32_t j = (uint32_t)t * 64u + (uint32_t)lane; qr[t] = qsrc[j < D ? j : 0u]; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the log's last word has reached L2 (it is read back past L1)
  for (uint32_t i0 = 0; i0 < n; i0 += WAVE) {                     // (uniform)
    const uint32_t i = i0 + (uint32_t)lane;
    const uint32_t id = ld_bypass_l1(cand + (i < n ? i : 0u));
    const uint8_t GAS* v = p.vec_base + (uint64_t)id * p.vec_stride;
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {                                 // the 64 dimensions whose query elements sit in qr[t]
      const uint32_t jt = (uint32_t)t * 64u;
      if (jt >= D) break;                                         // (uniform)
      const uint32_t dt = D - jt < 64u ? D - jt : 64u;
      for (uint32_t jl = 0; jl < dt; jl += 4u * RF) {             // (uniform) 4 RF dimensions = RF 16-byte loads in flight per lane and trip
        u32x4a w[RF];
#pragma unroll
        for (int u = 0; u < RF; ++u) {
          const uint32_t j = jl + 4u * (uint32_t)u;
          w[u] = *(const u32x4a GAS*)(v + 4u * (jt + (j < dt ? j : 0u)));       // (behind the vector's end: a piece again, never used)
        }
#pragma unroll
        for (int u = 0; u < RF; ++u) {
          const uint32_t j = jl + 4u * (uint32_t)u;
          if (j < dt) {                                           // (uniform)
            const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const float qv = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qr[t]), (int)(j + (uint32_t)d)));
              const float diff = __uint_as_float(ww[d]) - qv;     // :1294
              acc = __builtin_fmaf(diff, diff, acc);              // :1295, ascending dimension
            }
          }
        }
      }
    }
    if (i < n) e[i] = __float_as_uint(acc);
  }
  wave_sync();
  wave_topk(p, qabs, n, e, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side launcher helpers (per translation unit)
// ---------------------------------------------------------------------------------------------------------------------
#define HIP_TRY(x)                                                         \
  do {                                                                     \
    hipError_t _e = (x);                                                   \
    if (_e != hipSuccess) {                                                \
      bang_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
      return BANG_ERR_HIP;                                                 \
    }                                                                      \
  } while (0)

// Per-DEVICE launcher state: one process may drive several GPUs (one engine per device), and both the CU count and the
// dynamic-LDS attribute of a kernel instance belong to the device that is current at launch time.
#define BANG_MAX_DEVICES 64
static inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BANG_MAX_DEVICES) dev = 0;
  return dev;
}
static inline int num_cus() {
  static int cus[BANG_MAX_DEVICES] = {0};
  const int dev = current_device();
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    int n = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    cus[dev] = n > 0 ? n : 256;
  }
  return cus[dev];
}

// floats of the packed pivot table (multiple of 4: it is staged into LDS with 16-byte copies)
static inline uint32_t pivot_table_floats(uint32_t psz, uint32_t mp, uint32_t nhi) {
  if (psz == 2 && nhi != 0) return ((nhi * 512u + (mp - nhi) * 256u + 3u) & ~3u) + 4u;
  return mp * 256u * psz;
}

#endif
