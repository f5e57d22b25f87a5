// Primitives shared by the wave-specialised step kernels (mcpc_steps_ws2.h): progress counters in LDS, polled with
// bounded spins, and LDS-scoped hand-off fences.  A workgroup has G waves that only stream weight fragments and issue
// MFMAs and E waves that only run epilogues; every dependency between them is a monotonic counter
//   prog_e[k] / prog_g[k]  = table entries completed by E_k / G_k (absolute: step * n_entries + index + 1)
// polled by one ds_read per wait.  Spins are bounded (a broken schedule yields wrong results and a device error word,
// which the parity tests and mcpc_sync_check catch, never a hung GPU).
// (The first wave-specialised kernel, with 4 KiB LDS staging slots between the roles, lived here until it was superseded by
// the in-place variant: 109 vs 99 us per step at cfg-M.)
#pragma once

namespace mcpc {

#ifndef MCPC_WS_SLEEP
#define MCPC_WS_SLEEP 2
#endif
constexpr int kWsSpinLimit = 1 << 22;  // iterations (~0.1-0.2 us each): ~0.5 s, far beyond any legitimate wait (< 1 ms)

enum : int { PHF_WS_GEMM = 16, PHF_WS_EPI = 32 };   // which role has work in a table entry
enum : int { PHF_WS2_HANDOFF = 64 };                // in-place kernel: BWD entry without GEMM whose block (accb) still comes from G

// Everything the two roles hand to each other lives in LDS (global data written by an E wave -- x, spills, energies --
// is only re-read by the same lane or by later kernels), so the hand-off fences order LDS accesses only: a full
// workgroup-scope release would also drain every outstanding global store (`s_waitcnt vmcnt(0)`) at each publish.
#ifdef MCPC_EXP_FULLFENCE
#define MCPC_WS_FENCE(order_) __builtin_amdgcn_fence(order_, "workgroup")
#else
#define MCPC_WS_FENCE(order_) __builtin_amdgcn_fence(order_, "workgroup", "local")
#endif

__device__ __forceinline__ int ws_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ws_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// wait until a counter is >= need; a spin that runs out records the fact in *err (host: mcpc_sync_check)
// (a wave whose wait ran out once stops waiting altogether -- `dead` -- so a broken schedule drains in about one
// spin limit instead of one per remaining table entry).
// Shape matters here: a wait that is already satisfied -- most are -- must cost one LDS round trip and a compare.  Written as
// one `for` loop with the limit and the `dead` flag folded into its bounds, hipcc unrolled the poll eight times and chained
// ~60 scalar instructions behind it to reconstruct the spin count (~450 cycles per SATISFIED wait, three or four of them per
// table entry and pair): the satisfied case is now tested first, the polling loop is a cold, non-unrolled block.
__device__ __forceinline__ void ws_wait_one(const int* p, int need, int* err, int& dead) {
    if (__builtin_expect(ws_ld(p) < need, 0)) {
        if (!dead) {
            int spin = 0;
            bool ok = false;
#pragma clang loop unroll(disable)
            do {
                if (MCPC_WS_SLEEP > 0) __builtin_amdgcn_s_sleep(MCPC_WS_SLEEP);
                ok = ws_ld(p) >= need;
            } while (!ok && ++spin < kWsSpinLimit);
            if (!ok) { if ((threadIdx.x & 63) == 0) atomicOr(err, 2); dead = 1; }
        }
    }
    MCPC_WS_FENCE(__ATOMIC_ACQUIRE);
}
__device__ __forceinline__ void ws_publish(int* p, int v) {
    MCPC_WS_FENCE(__ATOMIC_RELEASE);     // LDS writes of this wave before the counter
    ws_st(p, v);
}

}  // namespace mcpc
