// bsk_probes.hpp — measurement probes of the step kernel, quarantined.
//
// The product library is built with NONE of this: every hook below is an empty inline function or a `false` constant, and
// the kernels carry no preprocessor conditionals for probes.  A probe library is a separate build,
//     make -C basilisk_env_amd/csrc probes          (-> ../variants/probe_*.so, picked up through BSKGPU_LIB)
// with -DBSK_PROBES=1 and exactly one selector:
//     BSK_PROBE_PAIR_WAIT     cycles the pair form's dynamics wave waits at its barriers, the other wave's chain time  (tools/pair_wait.py)
//     BSK_PROBE_PAIR_TIME     residency of every wave: cycles and hardware id                                            (tools/pair_time.py)
//     BSK_PROBE_PAIR_HWID     which SIMD hosts which wave of a pair                                                      (tools/pair_place.py)
//     BSK_PROBE_TRI_XCHG=1|2  the three-wave exchange: early consumes, re-reads, cycles re-reading (rotational / translational wave; tools/tri_wait.py)
//     BSK_PROBE_CHUNK=1..5    single-wave form: cycles of one part of every chunk of ticks beside the whole loop's (1: the chunk's head up to its
//                             first tick - FSW chain, flush, anchors; 2: the lone first tick + latch; 3: the FSW chain alone; 4: the power
//                             system's flush; 5: the third-body and density anchors; tools/chunk_probe.py)
//     BSK_PROBE_TRI_ROLE=1|2|3  three-wave form, ONE role per library (1 rotational, 2 FSW + environment, 3 translational wave): cycles in
//                             its tick loop, cycles of them at the workgroup barriers, cycles re-reading the exchange (dynamics halves) or
//                             in the FSW chain (environment wave); the workgroup's word of the debug buffer (tools/tri_roles.py)
//     BSK_PROBE_TRI_NOPUBLISH fault injection: the translational wave never publishes, so that its partner's poll times out
//                             (tests/test_gpu_tri.py: the handle's error word -> BSK_EHIP)
// A probe writes ONE 64-bit word per wave into the handle's debug buffer (bsk_debug_words), never into a result buffer.
// (Rounds 2-3 also carried timing-only ablations - BSK_ABLATE, BSK_PAIR_ABLATE, BSK_TRI_ABLATE - and A/B switches of settled
// decisions; their numbers are in profiles/r02, profiles/r03 with the commits that produced them, the code is gone.)
#pragma once
#include <hip/hip_runtime.h>

#if !defined(BSK_PROBES) || !BSK_PROBES
#if defined(BSK_PROBE_PAIR_WAIT) || defined(BSK_PROBE_PAIR_TIME) || defined(BSK_PROBE_PAIR_HWID) || defined(BSK_PROBE_TRI_XCHG) || defined(BSK_PROBE_TRI_NOPUBLISH) || defined(BSK_PROBE_CHUNK) || defined(BSK_PROBE_TRI_ROLE)
#error "a BSK_PROBE_* selector without -DBSK_PROBES=1: probes never ride along in a product build"
#endif
#endif

namespace bsk {
namespace probe {

#if defined(BSK_PROBES) && BSK_PROBES
#ifdef BSK_PROBE_PAIR_WAIT
constexpr bool PAIR_WAIT = true;
#else
constexpr bool PAIR_WAIT = false;
#endif
#ifdef BSK_PROBE_PAIR_TIME
constexpr bool PAIR_TIME = true;
#else
constexpr bool PAIR_TIME = false;
#endif
#ifdef BSK_PROBE_PAIR_HWID
constexpr bool PAIR_HWID = true;
#else
constexpr bool PAIR_HWID = false;
#endif
#ifdef BSK_PROBE_TRI_XCHG
constexpr int TRI_XCHG = BSK_PROBE_TRI_XCHG;
#else
constexpr int TRI_XCHG = 0;
#endif
#ifdef BSK_PROBE_CHUNK
constexpr int CHUNK = BSK_PROBE_CHUNK;
#else
constexpr int CHUNK = 0;
#endif
#ifdef BSK_PROBE_TRI_NOPUBLISH
constexpr bool TRI_NOPUBLISH = true;
#else
constexpr bool TRI_NOPUBLISH = false;
#endif
#ifdef BSK_PROBE_TRI_ROLE
constexpr bool TRI_ROLE = true;
constexpr int TRI_ROLE_WAVE = (BSK_PROBE_TRI_ROLE) - 1;     // the wave whose word is kept: 0 rotational, 1 FSW + environment, 2 translational
static_assert(TRI_ROLE_WAVE >= 0 && TRI_ROLE_WAVE <= 2, "BSK_PROBE_TRI_ROLE=1|2|3");
#else
constexpr bool TRI_ROLE = false;
constexpr int TRI_ROLE_WAVE = -1;
#endif
#else
constexpr bool PAIR_WAIT = false, PAIR_TIME = false, PAIR_HWID = false, TRI_NOPUBLISH = false, TRI_ROLE = false;
constexpr int TRI_XCHG = 0, CHUNK = 0, TRI_ROLE_WAVE = -1;
#endif
constexpr bool ANY = PAIR_WAIT || PAIR_TIME || PAIR_HWID || TRI_XCHG != 0 || CHUNK != 0 || TRI_ROLE;     // probes that emit a word per wave
constexpr bool XCH_STATS = TRI_XCHG != 0 || TRI_ROLE;      // the exchange keeps its miss / re-read / cycle counts
static_assert((int)PAIR_WAIT + (int)PAIR_TIME + (int)PAIR_HWID + (int)(TRI_XCHG != 0) + (int)TRI_NOPUBLISH + (int)(CHUNK != 0) + (int)TRI_ROLE <= 1, "one probe per library");
// three 20-bit fields of cycles / 64
__device__ __forceinline__ unsigned long long pack3(unsigned long long a, unsigned long long b, unsigned long long c) {
    return ((a >> 6) & 0xFFFFFull) | (((b >> 6) & 0xFFFFFull) << 20) | (((c >> 6) & 0xFFFFFull) << 40);
}

typedef unsigned long long Stamp;
// the cycle counter when probe ON is built in, 0 (and no instruction) otherwise
template <bool ON>
__device__ __forceinline__ Stamp stamp() {
    if constexpr (ON) return __builtin_readcyclecounter();
    else return 0ull;
}
template <bool ON>
__device__ __forceinline__ void since(unsigned long long& acc, Stamp t0) {
    if constexpr (ON) acc += __builtin_readcyclecounter() - t0;
}
__device__ __forceinline__ unsigned long long elapsed(Stamp t0) { return __builtin_readcyclecounter() - t0; }
// (xcc id << 16) | HW_ID: se, sh, cu, simd of the wave that asks
__device__ __forceinline__ unsigned long long hw_id() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (unsigned long long)((hw & 0xFFFFu) | ((xcc & 0xFu) << 16));
}

}  // namespace probe
}  // namespace bsk
