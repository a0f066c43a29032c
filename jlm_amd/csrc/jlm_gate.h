// jlm_gate.h -- what the two translation units of the fused LSTM step share (jlm_gate.hip: one tile per workgroup;
// jlm_gate_ws.hip: the W-stationary persistent kernel, compiled with its own code-generation flag -- __graft_entry__.py).
#pragma once
#include "jlm_common.h"
#include <type_traits>
#include <utility>

namespace jlm_gate {
struct GateXgArgs {
    const float *h; const float *c_in; float *h_out; float *c_out; int ld;
    float *h_f32;                                        // optional plain f32 copy of h' (untied models: T is the state itself)
    const int *rows, *prev, *word;
    const float *wt; const float *xg;
    int H; float descale, h_scale;
    int nrows; const int *ndev;
    int tiles_m, tiles_n;
    int cx;                                              // gate_ws_kernel: gate-column tiles per XCD (jlm_gate_ws.hip, tile map)
};
}  // namespace jlm_gate
using jlm_gate::GateXgArgs;

namespace {

constexpr int GT_BM = 160;                               // hypotheses per tile (5 blocks of 32)
constexpr int GT_BN = 128;                               // gate columns per tile (4 blocks of 32 = 32 units)


template <int N> using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void gate_for_each_ic(F &&f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }

}  // namespace

namespace jlm_gate {
// gate_ws_kernel<L, HF32> on tiles_n * Q workgroups (jlm_gate_ws.hip); L = ring stages ahead (3 or 7).  0, a hipError_t or -3.
int ws_launch(const GateXgArgs &a, int L, int Q, hipStream_t stream);
// gate_p2_kernel (jlm_gate_p2.hip): 128 x 256 tiles, a 2 x 2 register block per wave, persistent; H = 512 with a row list, no f32 copy of h'.
int p2_launch(const GateXgArgs &a, hipStream_t stream);
}
