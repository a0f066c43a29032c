// jlm_mx6.hip -- round 6: the vocabulary projection + log-sum-exp on mx6 rows: f16 hi.hi plus BOTH cross terms of the split product in
// one block-scaled FP6 matrix instruction per 32 k-values, accumulated into the same f32 accumulator (jlm_mx6_body.h has the scheme, the
// row format and the error figures).  Reference: project + softmax, decoder/model.py:141-193, 15-20.
// Launched by jlm_vocab_lse_mixed (jlm_mixed.hip) with its column cuts when every segment's rows are in this form (s8 = 0, ABI 11).
//
// Compiled with -fno-honor-nans -mno-amdgpu-ieee (__graft_entry__.py): the fold takes the running maximum straight from the matrix
// instruction's accumulator, and with IEEE semantics hipcc canonicalises every such value first (v_max_f32 x, x, x: one more VALU
// instruction per logit -- 742 instead of 266 v_max in the D-softmax* kernel).  A NaN logit is garbage either way.
#include "jlm_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "jlm_mx6_body.h"
using namespace jlm_mx;

namespace {

template <bool INLINE, bool XB, bool FR, int NB, int NS16>
struct Mx6Call {
    static __device__ __forceinline__ void run(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                               float2 *prow, unsigned char *smem) {
        mx6_body<NB, NS16, mx_blocks_per_tile(NB), XB, FR>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
    }
};
template <bool XB, bool FR, int NB, int NS16>
__device__ __noinline__ void mx6_body_outline(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                              float2 *prow, unsigned char *smem) {
    mx6_body<NB, NS16, mx_blocks_per_tile(NB), XB, FR>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
}
template <bool XB, bool FR, int NB, int NS16>
struct Mx6Call<false, XB, FR, NB, NS16> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                               float2 *prow, unsigned char *smem) {
        mx6_body_outline<XB, FR, NB, NS16>(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
    }
};
template <bool INLINE, bool XB, bool FR, int... SH>
struct Mx6Dispatch;
template <bool INLINE, bool XB, bool FR>
struct Mx6Dispatch<INLINE, XB, FR> {
    static __device__ __forceinline__ void run(const MxSeg &, int, int, int, int, int, const unsigned char *, int, float2 *, unsigned char *) {}
};
template <bool INLINE, bool XB, bool FR, int NB, int NS16, int... REST>
struct Mx6Dispatch<INLINE, XB, FR, NB, NS16, REST...> {
    static __device__ __forceinline__ void run(const MxSeg &sg, int ns16, int vt0, int vt1, int pt, int n_paths, const unsigned char *Tm, int ld_tm,
                                               float2 *prow, unsigned char *smem) {
        if (sg.nb == NB && ns16 == NS16) Mx6Call<INLINE, XB, FR, NB, NS16>::run(sg, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
        else Mx6Dispatch<INLINE, XB, FR, REST...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, smem);
    }
};
template <bool INLINE, bool XB, bool FR, int... SH>
__global__ __launch_bounds__(512, 1) void vocab_lse_mx6_kernel(MxArgs a, const unsigned char *__restrict__ Tm, int ld_tm, float2 *__restrict__ part,
                                                               int ld_part, int n_rows_max, const int *n_dev, int n_ptiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mx6_smem[];
    const int n_paths = n_dev ? min(*n_dev, n_rows_max) : n_rows_max;
    const int b = blockIdx.x;
    int p, pt;
    const int nb8 = (a.n_cols & ~7) * n_ptiles;
    if (b < nb8) { const int x = b & 7, jb = b >> 3; p = (jb / n_ptiles) * 8 + x; pt = jb % n_ptiles; }
    else { const int bb = b - nb8; p = (a.n_cols & ~7) + bb / n_ptiles; pt = bb % n_ptiles; }
    if (p >= a.n_cols || pt * 256 >= n_paths) return;
    for (int r = a.col_first[p]; r < a.col_first[p + 1]; ++r) {
        const MxSeg sg = a.seg[a.sub_seg[r]];
        const int vt0 = a.sub_t0[r], vt1 = a.sub_t1[r];
        float2 *prow = part + (size_t)r * ld_part;
        if (r != a.col_first[p]) __syncthreads();
        const int ns16 = XB ? 2 * sg.nb : (sg.k + 2 + 15) >> 4;
        Mx6Dispatch<INLINE, XB, FR, SH...>::run(sg, ns16, vt0, vt1, pt, n_paths, Tm, ld_tm, prow, mx6_smem);
    }
}
#define MX6_KERNEL_DSOFTMAX vocab_lse_mx6_kernel<true, false, false, 7, 13, 4, 7, 2, 4>
#define MX6_KERNEL_DSOFTMAX_FR vocab_lse_mx6_kernel<true, false, true, 7, 13, 4, 7, 2, 4>
#define MX6_KERNEL_GENERIC vocab_lse_mx6_kernel<false, false, false, 1, 1, 1, 2, 2, 3, 2, 4, 3, 5, 3, 6, 4, 7, 4, 8, 5, 9, 5, 10, 6, 11, 6, 12, 7, 13, 7, 14, 8, 15, 8, 16>
#define MX6_KERNEL_TIED vocab_lse_mx6_kernel<true, true, false, 8, 16>
#define MX6_KERNEL_TIED_FR vocab_lse_mx6_kernel<true, true, true, 8, 16>
#define MX6_KERNEL_GENERIC_XB vocab_lse_mx6_kernel<false, true, false, 2, 4, 4, 8, 6, 12, 8, 16>


}  // namespace

// the wide form (jlm_mx6w.hip: four waves of 64 rows): the shapes it hosts, its launch
bool jlm_mx6w_hosts(const MxArgs &a, bool xbias);
int jlm_mx6w_launch(const MxArgs &a, bool xbias, int fixed_ref, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev,
                    int n_ptiles, int lds, hipStream_t st);

#ifndef JLM_MX6_WIDE_DEFAULT
#define JLM_MX6_WIDE_DEFAULT -1
#endif

// which kernel: the D-softmax* shapes (inlined), any other bias-column shape, tied k = 256 (inlined), other external-bias shapes.
// JLM_MX6_WIDE: 1 the wide kernel for every shape it hosts, 0 the eight-wave kernel, -1 (default) where it measures faster -- the tied
// k = 256 shapes: 98-100 vs 107 us at V = 50 k / 2 560 rows, 1 384 vs 1 500 us at V = 100 k / 20 480 rows; the D-softmax* launch measures
// the same on both (60.5-60.9 vs 59.6-59.9 us) and stays on the eight-wave kernel (profiles/r06_g_mx6_wide.txt) -- as for the int8 planes.
// Returns 0, -3 (LDS grant) or a negative HIP error like its caller.
int jlm_mx6_launch(const MxArgs &a, bool xbias, int fixed_ref, const void *Tm, int ld_tm, float2 *part, int ld_part, int n_rows_max, const int *n_dev,
                   int n_ptiles, int lds, hipStream_t st) {
    static int wide = -1;
    if (wide < 0) { const char *e = getenv("JLM_MX6_WIDE"); wide = e ? atoi(e) : JLM_MX6_WIDE_DEFAULT; }
    if ((wide > 0 || (wide < 0 && xbias)) && jlm_mx6w_hosts(a, xbias))
        return jlm_mx6w_launch(a, xbias, fixed_ref, Tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles, lds, st);
    int which = xbias ? 2 : 0;
    for (int i = 0; i < a.n_segs; ++i) {
        const int nb = a.seg[i].nb, ns16 = (a.seg[i].k + 2 + 15) / 16;
        if (xbias) { if (nb != 8) which = 3; }
        else if (!((nb == 7 && ns16 == 13) || (nb == 4 && ns16 == 7) || (nb == 2 && ns16 == 4))) which = 1;
    }
    // fixed_ref: the forms without a running maximum exist for the inlined shapes (4: D-softmax*, 5: tied k = 256); the caller passes it
    // only for launches whose descale is 1 (the accumulators are base-2 logits)
    if (fixed_ref && which == 0) which = 4;
    if (fixed_ref && which == 2) which = 5;
    static JlmLdsGrant grant[6];
    const void *fns[6] = {reinterpret_cast<const void *>(MX6_KERNEL_DSOFTMAX), reinterpret_cast<const void *>(MX6_KERNEL_GENERIC),
                          reinterpret_cast<const void *>(MX6_KERNEL_TIED), reinterpret_cast<const void *>(MX6_KERNEL_GENERIC_XB),
                          reinterpret_cast<const void *>(MX6_KERNEL_DSOFTMAX_FR), reinterpret_cast<const void *>(MX6_KERNEL_TIED_FR)};
    if (int rc = jlm_grant_lds(grant[which], fns[which], lds)) return rc;
    const dim3 grid(a.n_cols * n_ptiles), block(512);
    const unsigned char *tm = reinterpret_cast<const unsigned char *>(Tm);
    switch (which) {
    case 0: hipLaunchKernelGGL(MX6_KERNEL_DSOFTMAX, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 1: hipLaunchKernelGGL(MX6_KERNEL_GENERIC, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 2: hipLaunchKernelGGL(MX6_KERNEL_TIED, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 3: hipLaunchKernelGGL(MX6_KERNEL_GENERIC_XB, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    case 4: hipLaunchKernelGGL(MX6_KERNEL_DSOFTMAX_FR, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    default: hipLaunchKernelGGL(MX6_KERNEL_TIED_FR, grid, block, lds, st, a, tm, ld_tm, part, ld_part, n_rows_max, n_dev, n_ptiles); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e - 100;
    return 0;
}
