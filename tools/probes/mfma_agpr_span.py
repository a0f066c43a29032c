"""Round 5 probe: does the rate of a pure MFMA stream depend on how many ACCUMULATION registers its B operands cycle through?
(the wide vocabulary kernel's k = 200 body keeps 216 of them and runs its matrix instructions at 52 cycles each, the k = 100 body
131 and 32 cycles).  Generates one kernel per variant -- an unrolled sequence of v_mfma (f16 / i8 alternating in pairs, four
accumulators, B operands a[4 q : 4 q + 3], q cycling over NQ quads) -- builds it with hipcc and times it with clock64.
Usage (GPU box): python tools/probes/mfma_agpr_span.py"""
import os, subprocess, sys, tempfile

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
%(KERNELS)s
int main() {
    unsigned long long *d; hipMalloc(&d, 4096 * 8);
    unsigned long long h[4096];
%(CALLS)s
    return 0;
}
'''
KERNEL = r'''
__global__ __launch_bounds__(256, 1) void k_%(name)s(unsigned long long *out, int iters) {
    f32x16 f0 = {0}, f1 = {0}; i32x16 i0 = {0}, i1 = {0};
    i32x4 fr = {(int)threadIdx.x, 1, 2, 3};
    asm volatile("s_nop 0" ::: %(clob)s);
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
%(body)s
            : "+v"(f0), "+v"(f1), "+v"(i0), "+v"(i1) : "v"(fr) : %(clob)s);
    }
    unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (iters < 0) { out[0] = (unsigned long long)f0[0] + f1[0] + i0[0] + i1[0]; }
}
'''
CALL = r'''
    { hipLaunchKernelGGL(k_%(name)s, dim3(%(grid)d), dim3(256), 0, 0, d, 10); hipDeviceSynchronize();
      hipLaunchKernelGGL(k_%(name)s, dim3(%(grid)d), dim3(256), 0, 0, d, %(iters)d); hipDeviceSynchronize();
      hipMemcpy(h, d, %(grid)d * 32, hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < %(grid)d * 4; ++i) s += h[i];
      printf("%(name)-28s %%6.1f cycles per matrix instruction (%(n)d per iteration, %(grid)d workgroups)\n", s / (%(grid)d * 4) / %(iters)d / %(n)d); }
'''


def variant(name, nq, bfile, grid=256, n=56, iters=2000, same_pair=False, stride=1):
    lines = []
    for m in range(n):
        q = ((m // 2 if same_pair else m) * stride) % nq
        breg = "%s[%d:%d]" % (bfile, 4 * q, 4 * q + 3)
        acc = m % 4
        if acc < 2:
            lines.append('            "v_mfma_f32_32x32x16_f16 %%%d, %%4, %s, %%%d\\n\\t"' % (acc, breg, acc))
        else:
            lines.append('            "v_mfma_i32_32x32x32_i8 %%%d, %%4, %s, %%%d\\n\\t"' % (acc, breg, acc))
    clob = ", ".join('"%s%d"' % (bfile, r) for r in range(4 * nq)) if bfile == "a" else ", ".join('"v%d"' % r for r in range(100, 100 + 4 * nq))
    if bfile == "v":
        lines = [l.replace("v[", "v[100+").replace("v[100+", "v[") for l in lines]
        lines = []
        for m in range(n):
            q = ((m // 2 if same_pair else m) * stride) % nq
            breg = "v[%d:%d]" % (100 + 4 * q, 100 + 4 * q + 3)
            acc = m % 4
            op = "v_mfma_f32_32x32x16_f16" if acc < 2 else "v_mfma_i32_32x32x32_i8"
            lines.append('            "%s %%%d, %%4, %s, %%%d\\n\\t"' % (op, acc, breg, acc))
    return (KERNEL % dict(name=name, body="\n".join(lines), clob=clob), CALL % dict(name=name, grid=grid, iters=iters, n=n))


def main():
    vs = [variant("agpr_27quads", 27, "a"), variant("agpr_54quads", 54, "a"), variant("agpr_32quads", 32, "a"), variant("agpr_40quads", 40, "a"),
          variant("agpr_54quads_1wg", 54, "a", grid=1), variant("agpr_54quads_pairs_share_b", 54, "a", same_pair=True),
          variant("agpr_1quad", 1, "a"), variant("vgpr_27quads", 27, "v"), variant("vgpr_36quads", 36, "v")]
    src = SRC % dict(KERNELS="".join(k for k, _ in vs), CALLS="".join(c for _, c in vs))
    d = tempfile.mkdtemp()
    p = os.path.join(d, "p.hip")
    open(p, "w").write(src)
    exe = os.path.join(d, "p")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, p])
    if "--build-only" in sys.argv:
        print("built", exe)
        return
    subprocess.check_call([exe])


main()
