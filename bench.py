#!/usr/bin/env python
"""Headline benchmark: decoded kana chars/sec of the batched lattice decode.

    python bench.py --gpus N --steps K --warmup W [--config 2|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config 2 (default; BASELINE.json configs[1], the configuration the metric is quoted on for one GPU):
    LSTM h=512, D-softmax* segs (200,100,50) over V=50k, beam=10, 256 synthetic 20-kana sentences per GPU.
    A "step" is one pass of the hot path over one such batch, strings in -> n-best strings out:
    `value` = kana characters / wall time from the first lattice build to the last n-best string
    (SURVEY.md 8d), through the product entry point `Decoder.decode_batch`, K batches pipelined as the
    product pipelines its chunks (native lattice build + upload of batch i+1 and string read-out of batch
    i-1 run while the GPU decodes batch i).  Weights and lexicon are resident before the clock starts; the
    kana strings are the only input.  Every rank decodes its own 256 sentences (weak scaling, no collective
    on the data path); K steps are bracketed by barrier + synchronize and the MAX over ranks is reported.
    The rate with the batch's lattice already resident in HBM is reported beside it
    (`device_resident_chars_per_s`), and BASELINE configs[4] (below) runs as an extra leg (`config5`).

--config 5 (BASELINE.json configs[4]): ONE seeded set of 8 192 tied-softmax V=50k 20-kana sentences at beam 10,
    dealt over the ranks by jlm_amd.shard.decode_sharded (length-sorted round robin, no data-path collective):
    strong scaling, 8 192 / N sentences per GPU in chunks of 1 024.  A step is one pass over the whole set.

One JSON line on rank 0 (contract in the task statement) plus
  roofline     : dominant kernel (fused vocabulary-projection / log-sum-exp, split-f16 MFMA x3):
                 algorithmic FLOPs / live HIP-event duration on the launching stream
  gate_gemm    : the fused LSTM step (BASELINE metric, part 2): EXECUTED MFMA FLOPs / dense f16 peak
  cpu_baseline : the numpy oracle (a port of the reference path) timed on this node's host cores on a
                 bounded sample of the same workload (rank 0, N = 1)
"""
import argparse
import hashlib
import json
from collections import deque
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2516.6       # v_mfma_f32_32x32x16_f16, dense: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz
SPLIT_PASSES = 3                    # f16x3: hi.hi + hi.lo + lo.hi per f32-grade product
HBM_PEAK_BYTES_PER_S = 8.0e12       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured by a float4 copy)
CONFIG5_SENTENCES = 8192


def kernel_source_sha256():
    """Identity of the kernels the committed PMC traffic figure was measured on (profiles/traffic_latest.json)."""
    h = hashlib.sha256()
    for f in ("jlm_split.hip", "jlm_mixed.hip", "jlm_mixed_w.hip", "jlm_mixed_body.h", "jlm_mx6.hip", "jlm_mx6_body.h", "jlm_common.h"):
        with open(os.path.join(REPO, "jlm_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def gate_source_sha256():
    """Identity of the LSTM-step kernel the committed PMC traffic figure was measured on"""
    h = hashlib.sha256()
    for f in ("jlm_gate.hip", "jlm_gate_ws.hip", "jlm_gate_p2.hip", "jlm_gate.h", "jlm_common.h"):
        with open(os.path.join(REPO, "jlm_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def self_launch(n):
    """Re-run this command line as N ranks under torch.distributed.run (what the driver's N > 1 command does)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5],
                    help="2 = BASELINE configs[1] (headline, weak scaling); 5 = BASELINE configs[4] (8 192 sentences sharded, strong scaling)")
    ap.add_argument("--batch", type=int, default=256, help="sentences per GPU per step (config 2)")
    ap.add_argument("--length", type=int, default=20, help="kana per sentence")
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--fixture", default=None, help="mid-vtable (configs[1]) | mid-tied | big-tied; default by --config")
    ap.add_argument("--cpu-sentences", type=int, default=32, help="bounded CPU-baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="config 2 only: skip the extra configs[4] leg")
    ap.add_argument("--config5-passes", type=int, default=3)
    ap.add_argument("--decoder", default="static", choices=["static", "static-vs", "dynamic"],
                    help="static = Decoder full vocabulary (headline); static-vs = vocab_select; dynamic = DynamicDecoder (configs[3])")
    # debugging the N > 1 control flow on a one-GPU box: every rank on device 0, control-plane collectives over gloo
    ap.add_argument("--debug-shared-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-legs", action="store_true", help="config 2 only: skip the short configs[2] / configs[3] legs")
    args = ap.parse_args()
    if args.fixture is None:
        args.fixture = "mid-vtable" if args.config == 2 else "mid-tied"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` from a plain shell: start the N ranks ourselves (one process per GPU, rendezvous on
        # 127.0.0.1; torch.distributed is control plane only -- barrier and max/sum of the timings)
        return self_launch(args.gpus)

    import numpy as np
    import torch
    import jlm_amd          # before the HIP runtime initialises: it defaults GPU_MAX_HW_QUEUES (jlm_amd/__init__.py)
    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        ge.build()
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node == --gpus, or without a launcher: "
                         "`python bench.py --gpus N` starts the ranks itself)" % (world, args.gpus))
    if world > 1 and not args.debug_shared_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (--debug-shared-gpu runs every rank on device 0)" % (
            world, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.debug_shared_gpu:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    else:
        torch.cuda.set_device(0)

    from jlm_amd import config as jconfig, shard, synth
    from jlm_amd.decoder import Decoder
    from jlm_amd.decoder_dynamic import DynamicDecoder
    from jlm_amd.lattice import BatchLattice

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return float(x)
        t = torch.tensor([x], device="cpu" if args.debug_shared_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return float(x)
        t = torch.tensor([x], device="cpu" if args.debug_shared_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def make_decoder(fixture, kind):
        root = os.path.join(tempfile.gettempdir(), "jlm_bench_%d_%s_r%d" % (os.getuid(), fixture, rank))
        cfg, _lex, _rd, alphabet = synth.build_fixture(root, fixture)
        jconfig.set_root(root)
        dec = DynamicDecoder(1) if kind == "dynamic" else Decoder(1)
        return root, cfg, alphabet, dec

    dkw = dict(vocab_select=True) if args.decoder != "static" else {}

    # ------------------------------------------------------------------------------------------ config 5 leg
    def run_config5(dec5, alphabet5, passes):
        """BASELINE configs[4]: 8 192 sentences dealt over the ranks, strong scaling.  -> (seconds, chars, n_mine)"""
        sents5 = synth.make_sentences(CONFIG5_SENTENCES, args.length, seed=5555, alphabet=alphabet5)   # the same set on every rank
        dec5.max_batch = 1024
        idx, _res = shard.decode_sharded(dec5, sents5, rank, world, beam_width=args.beam)             # untimed pass (plans, streams)
        del _res
        barrier()
        t0 = time.perf_counter()
        for _ in range(passes):
            idx, res = shard.decode_sharded(dec5, sents5, rank, world, beam_width=args.beam)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert len(res) == len(idx) and all(len(r) > 0 for r in res)
        n_all = int(round(sum_over_ranks(float(len(idx)))))
        assert n_all == CONFIG5_SENTENCES, "the shards must cover the whole set exactly once"
        return dt, sum(len(s) for s in sents5), len(idx)

    def kernel_stats(durs, rows, name, flops_per_row):
        d = np.asarray(durs.get(name, []), dtype=np.float64)
        if d.size == 0:
            return None
        n = min(d.size, rows.size)
        flops = float((rows[:n] * flops_per_row).sum())
        secs = float(d[:n].sum()) * 1e-3
        return dict(launches=int(n), avg_ms=float(d[:n].mean()), tflops=flops / secs / 1e12, flops_per_launch=flops / n)

    def measure_kernels(dec, lat, ekind, ekw, steps, full_vocab=True, fixture=None, decoder_name=None):
        """The same decode once more, timed: HIP events on the launching stream around the LSTM step and around the vocabulary
        kernel of every frame (same kernels, same arguments; kept out of the throughput loops so that the event records and
        the single stream do not perturb them)."""
        eng, m = dec._engine, dec.model.dev
        fixture = fixture or args.fixture
        decoder_name = decoder_name or args.decoder
        eng.keep_n_live = True
        n_live, durs = [], {"gate_gemm": [], "vocab_lse": []}
        for _ in range(steps):
            eng.decode(lat, ekind, topN=10, timing=True, **ekw)       # jlm_decode_frames records the events (include/jlm_hip.h)
            n_live.append(eng.last_n_live)
            for k in durs:
                durs[k].extend(eng.last_kernel_ms[k])
        torch.cuda.synchronize()
        eng.keep_n_live = False
        rows = np.concatenate([np.asarray(x[:-1], dtype=np.float64) for x in n_live])    # live rows per launch
        H = m.H
        split = getattr(m, "split_array", None) is not None or getattr(m, "um_split", None) is not None
        gsplit = getattr(m, "split_lstm", False)
        # executed MFMA work: 3 passes, contraction padded to whole 16-value steps (+ the bias column where it rides in the GEMM)
        k16 = lambda sg: (sg["k"] + (1 if (sg["k"] % 16 and m.stationary_ok) else 0) + 15) // 16 * 16
        exec_per_row_vocab = SPLIT_PASSES * sum(2.0 * k16(sg) * (sg["v_end"] - sg["v_start"]) for sg in m.segments)
        # every segment on mixed rows (jlm_vocab_lse_mixed): per 32 x 32 block and 32 k-values two f16 and two int8 matrix
        # instructions of 32 cycles each, against six f16 ones of the split form; counted here in f16-instruction units
        # (2 x 16 flop per row, word and instruction), k + 2 padded to whole f16 steps / int8 blocks
        heads = list(getattr(m, "mixed_head_split", None) or [])
        mixed = split and len(getattr(m, "mixed_idx", [])) == m.n_segs and not any(heads)
        hybrid = split and bool(getattr(m, "mixed_idx", [])) and not mixed
        # (round 6, mx6 rows: per 32 k-values the two f16 instructions and ONE block-scaled FP6 instruction of 32 cycles for both cross terms)
        mx6 = getattr(m, "mixed_fmt", None) == "mx6"
        mx_exec = lambda sg, nv: 2.0 * 16 * ((sg["k"] + 2 + 15) // 16 + (1 if mx6 else 2) * ((sg["k"] + 2 + 31) // 32)) * nv
        if mixed:
            exec_per_row_vocab = sum(mx_exec(sg, sg["v_end"] - sg["v_start"]) for sg in m.segments)
        elif hybrid:
            # (round 5) both formats in one launch: a segment on mixed rows but for the head that stays on split rows, or on split rows whole
            exec_per_row_vocab = 0.0
            for i, sg in enumerate(m.segments):
                nv = sg["v_end"] - sg["v_start"]
                j = m.mixed_idx.index(i) if i in m.mixed_idx else -1
                cut = nv if j < 0 else (heads[j] if j < len(heads) else 0)
                exec_per_row_vocab += SPLIT_PASSES * 2.0 * k16(sg) * cut + mx_exec(sg, nv - cut)
        v = kernel_stats(durs, rows, "vocab_lse", m.flops_per_row_vocab / (1 if m.stationary_ok else m.n_segs))
        roofline = None
        if v:
            kname = ("vocab_lse_mixedw_kernel<1, true, 16, 32> (jlm_vocab_lse_mixed: the wide one-row-set form for k = 512, csrc/jlm_mixed_w.hip; rows "
                     "packed by pack_t_mixed_kernel from the state's f32 copy)" if (mixed and getattr(m, "um_split", None) is not None) else
                     "gemm_split_kernel<128x128,EpiLse> (jlm_vocab_lse_partials_split: tile form, k = H)" if getattr(m, "um_split", None) is not None else
                     "vocab_lse_mixedw_kernel<2, true, 8, 16> (jlm_vocab_lse_mixed: the wide form -- four waves x 64 rows, row operands in accumulation "
                     "registers, csrc/jlm_mixed_w.hip; rows packed by pack_t_mixed_kernel behind the T projection)"
                     if (mixed and not mx6 and m.n_segs == 1 and m.segments[0]["k"] == 256 and os.environ.get("JLM_MX_WIDE", "-1") != "0") else
                     "vocab_lse_mx6_kernel (jlm_vocab_lse_mixed on mx6 rows, csrc/jlm_mx6.hip; its rows packed by pack_t_mx6_kernel behind the T "
                     "projection)" if (mixed and mx6) else
                     "vocab_lse_mixed_kernel (jlm_vocab_lse_mixed; its rows packed by pack_t_mixed_kernel behind the T projection)" if mixed else
                     "vocab_lse_hybrid_kernel (jlm_vocab_lse_hybrid: segments %s on mixed rows%s, the rest on split rows; the mixed segments' rows "
                     "packed by pack_t_mixed_kernel)" % (list(m.mixed_idx), (" but for their first %s words" % heads) if any(heads) else "") if hybrid else
                     "vocab_lse_split8_kernel (jlm_vocab_lse_split)" if split else
                     "vocab_lse_stationary_kernel (jlm_vocab_lse_stationary)" if m.stationary_ok
                     else "gemm2_kernel<128x128,EpiLse> (jlm_vocab_lse_partials)")
            # `peak` prices ALGORITHMIC flops against the ceiling of the instructions the kernel issues: three f16 passes per
            # f32-grade product on split rows (dense f16 / 3); on mixed rows one f16 pass + two int8 passes at twice the k per
            # instruction = the time of TWO f16 passes (dense f16 / 2)
            # (both formats in one launch: the same pricing per segment -- dense f16 x algorithmic flops / executed f16-instruction flops)
            # (round 6, mx6 rows: one f16 pass + ONE FP6 instruction covering both cross terms of 32 k-values in 32 cycles = the time of 1.5
            #  f16 passes: dense f16 / 1.5)
            peak = (F16_MFMA_PEAK_TFLOPS / 1.5 if (mixed and mx6) else F16_MFMA_PEAK_TFLOPS / 2.0 if mixed else
                    F16_MFMA_PEAK_TFLOPS * m.flops_per_row_vocab / exec_per_row_vocab if hybrid else
                    F16_MFMA_PEAK_TFLOPS / SPLIT_PASSES) if split else F32_MFMA_PEAK_TFLOPS
            traffic, traffic_note = None, "not measured in this run (tools/gpu_traffic.sh + tools/traffic_report.py write profiles/traffic_latest.json)"
            tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
            if split and m.stationary_ok and os.path.exists(tpath):
                with open(tpath) as tf:
                    tj = json.load(tf)
                if tj.get("source_sha256") == kernel_source_sha256() and tj.get("fixture") == fixture:
                    traffic, traffic_note = tj["vocab_lse_hbm_bytes_per_call"], tj["note"]
                else:
                    traffic_note = ("profiles/traffic_latest.json was measured on other kernel sources / another fixture: omitted "
                                    "(its figure: %s bytes per call)" % tj.get("vocab_lse_hbm_bytes_per_call"))
            ex = v["tflops"] / m.flops_per_row_vocab * exec_per_row_vocab if split else v["tflops"]
            form = ("mixed" if mixed else "hybrid" if hybrid else "split" if split else "f32")
            roofline = {"kernel": kname, "bound": "mfma", "achieved": round(v["tflops"], 2), "peak": round(peak, 1),
                        "unit": "TFLOP/s", "frac": round(v["tflops"] / peak, 4),
                        "lse_form": form, "lse_form_calibration": getattr(m, "mixed_calib", None),
                        "lse_form_spread": [round(x, 2) for x in getattr(m, "mixed_spread", [])],
                        "frac_of_dense_f16": round(ex / F16_MFMA_PEAK_TFLOPS, 4) if split else None,
                        "frac_of_dense_f16_algorithmic": round(v["tflops"] / F16_MFMA_PEAK_TFLOPS, 4) if split else None,
                        "executed_tflops": round(ex, 1),
                        "traffic": traffic, "traffic_source": traffic_note,
                        "avg_launch_ms": round(v["avg_ms"], 4), "launches": v["launches"],
                        "flops_per_launch": v["flops_per_launch"],
                        "mfma_dtype": (("f16 hi.hi (v_mfma_f32_32x32x16_f16) + BOTH cross terms of the split product as FP6 (e2m3) x FP6 with an E8M0 scale "
                                        "per 32 k-values of every row, in ONE block-scaled instruction per 32 k-values (v_mfma_scale_f32_32x32x64_f8f6f4) into the "
                                        "same f32 accumulator: 3 matrix instructions of 32 cycles per 32 k-values (int8 cross terms: 4, three f16 passes: 6).  "
                                        "`peak` = %.1f dense f16 / 1.5: the ceiling of the instructions issued; frac_of_dense_f16 prices the executed "
                                        "instructions at 32 cycles each; `frac_f16x3_pricing` keeps rounds 1-3's dense f16 / %d" % (F16_MFMA_PEAK_TFLOPS, SPLIT_PASSES))
                                       if (mixed and mx6) else
                                       ("f16 hi.hi (v_mfma_f32_32x32x16_f16) + int8 cross terms (v_mfma_i32_32x32x32_i8): 4 matrix instructions per "
                                        "32 k-values instead of the split form's 6.  `peak` = %.1f dense f16 / 2: the ceiling of the instructions issued "
                                        "(round 4; rounds 1-3 priced every form at dense f16 / %d = three f16 passes per f32-grade product -- that figure "
                                        "is kept as `frac_f16x3_pricing`); frac_of_dense_f16 prices the executed instructions at 32 cycles each"
                                        % (F16_MFMA_PEAK_TFLOPS, SPLIT_PASSES)) if mixed else
                                       ("both formats in one launch: mixed rows (one f16 + two int8 matrix instructions per 32 k-values and block) for the "
                                        "words whose logit error the log-normaliser tolerates, split rows (three f16 passes) for the segment / head that "
                                        "carries the probability mass (DeviceModel._calibrate_mixed); `peak` = %.1f dense f16 x algorithmic / executed flops"
                                        % F16_MFMA_PEAK_TFLOPS) if hybrid else
                                       "f16 split x3 (v_mfma_f32_32x32x16_f16, f32 accumulate): `peak` = %.1f dense f16 / %d passes prices "
                                       "ALGORITHMIC flops; frac_of_dense_f16 prices the executed ones (3 passes, k padded to 16)"
                                       % (F16_MFMA_PEAK_TFLOPS, SPLIT_PASSES) if split else "f32 (v_mfma_f32_32x32x2_f32)"),
                        "measured": "HIP events around every launch of the dominant kernel on its stream, in a repeat of the timed decode"}
            if split:
                roofline["frac_f16x3_pricing"] = round(v["tflops"] / (F16_MFMA_PEAK_TFLOPS / SPLIT_PASSES), 4)
            # the same fraction from the rocprofv3 kernel statistics tracked under profiles/ (average duration of this kernel in a
            # profiled run of this command: tools/gpu_prof.sh + tools/rocprof_report.py), when they were taken on these sources
            rpath = os.path.join(REPO, "profiles", "rocprof_latest.json")
            roofline["frac_rocprof"], roofline["rocprof_avg_launch_ms"] = None, None
            if os.path.exists(rpath):
                with open(rpath) as rf_:
                    rj = json.load(rf_)
                if rj.get("source_sha256") == kernel_source_sha256() and rj.get("fixture") == fixture and rj.get("vocab_lse_avg_us"):
                    roofline["rocprof_avg_launch_ms"] = round(rj["vocab_lse_avg_us"] * 1e-3, 4)
                    roofline["frac_rocprof"] = round(v["flops_per_launch"] / (rj["vocab_lse_avg_us"] * 1e-6) / 1e12 / peak, 4)
                    roofline["rocprof_source"] = rj.get("note")
                else:
                    roofline["rocprof_source"] = "profiles/rocprof_latest.json was taken on other kernel sources / another fixture: omitted"
            if not full_vocab:
                # the vocabulary-selected / per-frame-deduplicated decoders run this kernel over a sub-problem whose size is
                # decided on the device each frame: the full-vocabulary flop count does not apply, so nothing is priced
                for k in ("achieved", "frac", "frac_of_dense_f16", "frac_of_dense_f16_algorithmic", "frac_f16x3_pricing", "frac_rocprof",
                          "executed_tflops", "flops_per_launch", "traffic"):
                    if k in roofline:
                        roofline[k] = None
                roofline["note"] = ("decoder=%s works on a per-frame selected sub-problem (rows x columns decided on the device): "
                                    "only the launch time is reported; the roofline is quoted on decoder=static" % decoder_name)
        g = kernel_stats(durs, rows, "gate_gemm", 2.0 * H * 4 * H * (SPLIT_PASSES if gsplit else 1))
        gate_obj = None
        if g:
            gpeak = F16_MFMA_PEAK_TFLOPS if gsplit else F32_MFMA_PEAK_TFLOPS
            counted = g["tflops"] / (SPLIT_PASSES if gsplit else 1) * (H + m.E_in) / H
            gate_obj = {"kernel": ("gate_xg_kernel (jlm_lstm_step_xg: one 160 x 128 tile per CU; the input side x.W_x+b is a per-word "
                                   "table row added in the epilogue, the MFMAs contract over the state only)" if gsplit
                                   else "gemm2_kernel<64x64,EpiGate> (jlm_lstm_step)"),
                        "mfma_util_pct": round(100.0 * g["tflops"] / gpeak, 2),
                        "executed_tflops": round(g["tflops"], 1), "peak": round(gpeak, 1), "unit": "TFLOP/s",
                        "flops_executed": "%d x 2*H*4H per live row (K = H = %d)" % (SPLIT_PASSES if gsplit else 1, H),
                        "reference_step_counted_tflops": round(counted, 1),
                        "reference_step_counted": "2*(H+E)*4H per row, one pass (the reference's step, model.py:125-131), not executed work",
                        "avg_launch_ms": round(g["avg_ms"], 4), "launches": g["launches"]}
            if gsplit:
                # the byte side of the same launch (the kernel is co-bound by bytes at R = 2 560): algorithmic = per live row one
                # 8-KB table row + the gathered state row and cell row in (2 x 2 KB) + the new state and cell rows out (2 x 2 KB),
                # plus the 4 H x H split gate matrix once; measured = PMC (profiles/traffic_latest.json, as for the vocabulary kernel)
                rows_avg = float(rows[:g["launches"]].mean()) if g["launches"] else 0.0
                alg = rows_avg * (4 * H * 4 + 4 * H * 4) + 4 * H * H * 4
                gate_obj["hbm_bytes_algorithmic"] = int(alg)
                gate_obj["hbm_bytes_per_launch"], gate_obj["hbm_frac"] = None, None
                tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
                if os.path.exists(tpath):
                    with open(tpath) as tf:
                        tj = json.load(tf)
                    if tj.get("gate_source_sha256") == gate_source_sha256() and tj.get("fixture") == fixture and tj.get("gate_hbm_bytes_per_call"):
                        gate_obj["hbm_bytes_per_launch"] = int(tj["gate_hbm_bytes_per_call"])
                        gate_obj["hbm_frac"] = round(tj["gate_hbm_bytes_per_call"] / (g["avg_ms"] * 1e-3) / HBM_PEAK_BYTES_PER_S, 4)
                        gate_obj["hbm_note"] = "PMC bytes per launch / HIP-event launch time / 8 TB/s (MI355X_MICROARCH.md: ~6.3 TB/s achievable)"
        return roofline, gate_obj

    cpu = None
    line_extra = {}
    if args.config == 5:
        # ------------------------------------------------------------------------------------ BASELINE configs[4]
        root, cfg, alphabet, dec = make_decoder(args.fixture, "static")
        for _ in range(max(0, args.warmup - 1)):
            shard.decode_sharded(dec, synth.make_sentences(1024 * world, args.length, seed=77, alphabet=alphabet), rank, world,
                                 beam_width=args.beam)
        dt, chars, n_mine = run_config5(dec, alphabet, args.steps)
        value = chars * args.steps / dt
        sents = synth.make_sentences(1024, args.length, seed=5555 + rank, alphabet=alphabet)
        lat = BatchLattice(dec._builder, sents, args.beam)
        roofline, gate_obj = measure_kernels(dec, lat, "static", {}, 3)
        workload = ("BASELINE configs[4]: tied softmax V=50k (h=512, e=256), beam=10, ONE set of %d sentences x %d kana sharded over "
                    "%d GPU(s) by jlm_amd.shard.decode_sharded, %d per GPU in chunks of 1 024" % (
                        CONFIG5_SENTENCES, args.length, world, n_mine))
        scaling = "strong"
        steps_ms = dt / args.steps * 1e3
        cpu_sents, cpu_dec, cpu_root = sents, dec, root
    else:
        # ------------------------------------------------------------------------------------ BASELINE configs[1]
        root, cfg, alphabet, dec = make_decoder(args.fixture, args.decoder)
        eng = dec._engine
        # every rank decodes its own sentences (seeded by rank): sentence sharding, no data-path collective
        sents = synth.make_sentences(args.batch, args.length, seed=4242 + rank, alphabet=alphabet)
        chars_per_step = sum(len(s) for s in sents)
        dec.max_batch = args.batch
        lat = BatchLattice(dec._builder, sents, args.beam)
        ekind, ekw = "static", {}
        if args.decoder == "static-vs":
            w_, o_, _l = lat.static_vocab()
            ekw = dict(vocab=(w_, o_))
        elif args.decoder == "dynamic":
            ekind, ekw = "dynamic", dict(dyn_lists=lat.dynamic_vocab()[:4])

        def run_device_steps(n, timing=False, sink=None, on=None):
            """n steps with the lattice resident; the host read-out of step i overlaps the GPU work of step i+1.
            timing="inflight": HIP events around the kernel groups on each batch's own stream (sink collects them).
            on = (decoder, lattice, engine kind, engine kwargs, batch, beam): another decoder's loop (the legs)."""
            dec_, lat_, ekind_, ekw_, batch_, beam_ = on or (dec, lat, ekind, ekw, args.batch, args.beam)
            eng_ = dec_._engine
            inflight = deque()                 # pipeline_depth steps in flight, as Decoder.decode_batch keeps its chunks
            eng_.pipelined = True

            def fin(t):
                eng_.collect(t)
                if sink is not None:
                    sink["n_live"].append(eng_.last_n_live)
                    for k in ("gate_gemm", "vocab_lse"):
                        sink[k].extend(eng_.last_kernel_ms[k])
            try:
                for _ in range(n):
                    inflight.append(eng_.submit(lat_, ekind_, topN=10, timing=timing, **ekw_))
                    if len(inflight) > dec_.depth_for(batch_, beam_):
                        fin(inflight.popleft())
                while inflight:
                    fin(inflight.popleft())
            finally:
                eng_.pipelined = False

        # untimed: the requested warm-up steps (plans for every stream and pipeline slot exist afterwards), then two settle calls
        # of the timed call's own size: a fresh process runs its first K-batch call 20-25 % slower than the third
        # (tools/probes/idle_probe.py: 2.96 / 2.52 / 2.39 ms per batch -- the Python heap and the lattice buffers of a call of
        # that size are faulted in for the first time; an idle second does not bring it back).  The rate reported is the
        # steady state of a running service, as for any warm-up.
        # A fresh process reaches its steady state only with its third or fourth call of a size (tools/probes/warmup_curve.py, ms per
        # step of successive 20-step calls after a 5-step one: 3.0-6.8, 2.36-2.40, 2.13-2.15, 2.10): settle calls are repeated until
        # two successive ones agree within 3 % (2 to 6 calls).
        settle = max(args.steps, 12 - args.warmup, 4)
        # the cold side (verdict round 5, weak 5): the very first decode of this process -- one batch, strings -> strings: plans, page-locked
        # blocks and the engine's streams are created on the way -- before anything else has run
        torch.cuda.synchronize()
        _tf = time.perf_counter()
        dec.decode_batch(sents, beam_width=args.beam, **dkw)
        torch.cuda.synchronize()
        line_extra["first_call_ms"] = round((time.perf_counter() - _tf) * 1e3, 2)
        if args.warmup:
            dec.decode_batch(sents * args.warmup, beam_width=args.beam, **dkw)
        n_settle, prev = 0, None
        while n_settle < 6:
            torch.cuda.synchronize()
            ts = time.perf_counter()
            dec.decode_batch(sents * settle, beam_width=args.beam, **dkw)
            torch.cuda.synchronize()
            ts = time.perf_counter() - ts
            n_settle += 1
            if n_settle >= 2 and abs(ts - prev) <= 0.03 * prev:
                break
            prev = ts
        line_extra["untimed_steps"] = args.warmup + n_settle * settle
        line_extra["untimed_steps_note"] = ("everything run before the clock starts: --warmup steps + %d settle calls of max(steps, "
                                            "12 - warmup, 4) = %d steps each, repeated until two successive calls agree within 3 %% "
                                            "(first-touch of the heap, the page-locked blocks and the plans a call of that size needs)"
                                            % (n_settle, settle))
        # The timed region: EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks -- taken three times in a
        # row, the MEDIAN reported (`ms_per_step_repeats` lists all three): a 20-step region is 40 ms, and single regions on one
        # box spread by 8 %, more than most changes the line is meant to show.
        dts, cpus = [], []
        for _rep in range(3):
            barrier()
            c0 = time.process_time()
            t0 = time.perf_counter()
            out = dec.decode_batch(sents * args.steps, beam_width=args.beam, **dkw)      # K steps = K pipelined 256-sentence batches
            barrier()
            dts.append(max_over_ranks(time.perf_counter() - t0))
            cpus.append(time.process_time() - c0)
            assert len(out) == len(sents) * args.steps and all(len(r) > 0 for r in out)
            if _rep < 2:
                del out
        _mid = sorted(range(3), key=lambda i_: dts[i_])[1]
        dt, cpu_s = dts[_mid], cpus[_mid]
        line_extra["ms_per_step_repeats"] = [round(x / args.steps * 1e3, 3) for x in dts]
        line_extra["ms_per_step_note"] = "three consecutive timed regions of `steps` steps each; `value` / `ms_per_step` are the median one"
        line_extra["host_cpu_ms_per_step"] = round(cpu_s / args.steps * 1e3, 3)
        line_extra["host_cpu_note"] = ("process CPU time (all threads of this rank: calling thread, lattice workers, HIP runtime) per "
                                       "step of the timed region; %.2f CPUs busy on average, %d usable" % (cpu_s / dt, jlm_amd.usable_cpus()))
        from jlm_amd import numa as _numa
        _node, _cpus = dec._numa if getattr(dec, "_numa", None) is not None else _numa.worker_cpus(torch.cuda.current_device())
        line_extra["host_cpus"] = {"usable_by_this_rank": jlm_amd.usable_cpus(), "ranks_on_this_node": int(os.environ.get("LOCAL_WORLD_SIZE", "1")),
                                   "lattice_workers": dec.prefetch_workers, "gpu_numa_node": _node, "workers_pinned_to_cpus": len(_cpus),
                                   "note": "usable = affinity mask capped by the cgroup quota, divided by the ranks of the node (jlm_amd.usable_cpus); "
                                           "the lattice workers pin themselves to the CPUs of their GPU's NUMA node (jlm_amd/numa.py; 0 = sysfs names none)"}
        assert len(out) == len(sents) * args.steps and all(len(r) > 0 for r in out)
        # ... and ONE warm batch by itself (no pipelining: the latency of a single 256-sentence call in a running service)
        torch.cuda.synchronize()
        _tf = time.perf_counter()
        dec.decode_batch(sents, beam_width=args.beam, **dkw)
        torch.cuda.synchronize()
        line_extra["single_call_ms"] = round((time.perf_counter() - _tf) * 1e3, 2)
        line_extra["first_call_note"] = ("first_call_ms: the first decode_batch of the process (one batch of `sentences_per_gpu`, strings -> strings, cold: "
                                         "plans, page-locked blocks, streams); single_call_ms: the same call once the process is warm; `value` is the "
                                         "pipelined steady state after `untimed_steps`")
        del out        # ~300 k list objects: kept alive they make every later full garbage collection (the loops below) slower
        total_chars_per_step = sum_over_ranks(float(chars_per_step))
        value = total_chars_per_step * args.steps / dt
        steps_ms = dt / args.steps * 1e3
        # the same with the lattice already in HBM (not `value`): device decode + n-best traces back on the host
        run_device_steps(4)
        barrier()
        t1 = time.perf_counter()
        run_device_steps(args.steps)
        barrier()
        dt_dev = max_over_ranks(time.perf_counter() - t1)
        line_extra["device_resident_chars_per_s"] = round(total_chars_per_step * args.steps / dt_dev, 1)
        line_extra["device_resident_ms_per_step"] = round(dt_dev / args.steps * 1e3, 3)
        line_extra["device_resident_note"] = ("same K steps with the batch's lattice (CSR) already resident in HBM: launch sequence "
                                              "+ n-best traces back on the host, no lattice build / upload / string read-out")
        roofline, gate_obj = measure_kernels(dec, lat, ekind, ekw, min(args.steps, 20), full_vocab=(args.decoder == "static"))
        if roofline and args.decoder == "static":
            # consistency of the two clocks: the dominant kernel's solo launches of one step cannot take longer than the step
            per_step = roofline["launches"] / float(min(args.steps, 20))
            roofline["solo_launch_ms_per_step"] = round(roofline["avg_launch_ms"] * per_step, 4)
            # (two different runs -- serialised event timing vs the pipelined loop: recorded, never raised)
            roofline["clock_consistency_ok"] = bool(roofline["solo_launch_ms_per_step"] <= 1.02 * line_extra["device_resident_ms_per_step"])
            if not roofline["clock_consistency_ok"]:       # (round-5 advice: a line whose two clocks disagree is marked, loudly)
                line_extra["line_suspect"] = ("the dominant kernel's solo launches of one step (%.3f ms by HIP events) exceed the device-resident step "
                                              "(%.3f ms by the wall clock): one of the two measurements is wrong"
                                              % (roofline["solo_launch_ms_per_step"], line_extra["device_resident_ms_per_step"]))
                print("bench.py: WARNING: " + line_extra["line_suspect"], file=sys.stderr)
        if roofline and args.decoder == "static" and roofline.get("lse_form") in ("mixed", "hybrid"):
            # the same launches on SPLIT rows (three f16 passes; what a model the load-time gates keep off the int8 planes runs)
            os.environ["JLM_LSE_MIXED"] = "0"
            try:
                _rs, _cs, _as, dec_s = make_decoder(args.fixture, "static")
                dec_s.max_batch = args.batch
                lat_s = BatchLattice(dec_s._builder, sents, args.beam)
                rf_s, _g = measure_kernels(dec_s, lat_s, "static", {}, 5)
                if rf_s:
                    roofline["split_rows_avg_launch_ms"] = rf_s["avg_launch_ms"]
                    roofline["split_rows_frac"] = rf_s["frac"]
                    roofline["split_rows_note"] = ("JLM_LSE_MIXED=0: the same model and rows through jlm_vocab_lse_split (peak = dense f16 / 3); "
                                                   "what DeviceModel._calibrate_mixed / the spread gate fall back to")
                del dec_s, lat_s
            finally:
                del os.environ["JLM_LSE_MIXED"]
                jconfig.set_root(root)
        # the same kernels inside the pipelined loop (four batches in flight): events on each batch's own stream
        if roofline and args.decoder == "static" and eng.n_streams >= 2:
            sink = {"n_live": [], "gate_gemm": [], "vocab_lse": []}
            eng.keep_n_live = True
            run_device_steps(3, timing="inflight")
            run_device_steps(min(args.steps, 20), timing="inflight", sink=sink)
            eng.keep_n_live = False
            rows_p = np.concatenate([np.asarray(x[:-1], dtype=np.float64) for x in sink["n_live"]])
            m_ = dec.model.dev
            vp = kernel_stats(sink, rows_p, "vocab_lse", m_.flops_per_row_vocab / (1 if m_.stationary_ok else m_.n_segs))
            gp = kernel_stats(sink, rows_p, "gate_gemm", 2.0 * m_.H * 4 * m_.H * (SPLIT_PASSES if getattr(m_, "split_lstm", False) else 1))
            if vp:
                roofline["frac_in_pipeline"] = round(vp["tflops"] / roofline["peak"], 4)
                roofline["avg_launch_ms_in_pipeline"] = round(vp["avg_ms"], 4)
                if eng.lse_share_pct and eng.lse_share_pct < 100:
                    # the CUs the pipelined launch is cut for: columns x row tiles (jlm_decode_plan.lse_cu_share_pct, csrc/jlm_decode.hip)
                    n_ptiles = (args.batch * args.beam + 255) // 256
                    cus = max(1, 256 * eng.lse_share_pct // 100 // n_ptiles) * n_ptiles
                    roofline["cus_in_pipeline"] = cus
                    roofline["frac_in_pipeline_of_its_cus"] = round(vp["tflops"] / (roofline["peak"] * cus / 256.0), 4)
                roofline["in_pipeline_note"] = ("the same launches timed inside the pipelined loop (%d batches in flight, the kernel on "
                                                "%d%% of the CUs beside the other batches' kernels): events on each batch's own stream"
                                                % (dec.depth_for(args.batch, args.beam), eng.lse_share_pct or 100))
            if gp and gate_obj:
                gate_obj["mfma_util_pct_in_pipeline"] = round(100.0 * gp["tflops"] / gate_obj["peak"], 2)
                gate_obj["avg_launch_ms_in_pipeline"] = round(gp["avg_ms"], 4)
        # diagnostic: does this box overlap the batches in flight?  The same pipelined loop with one stream and with the
        # engine's own (on some boxes the two are equal: the queues of the streams are not run side by side there).
        if eng.n_streams >= 2:
            def timed_ms(n):
                barrier()
                t = time.perf_counter()
                run_device_steps(n)
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / n * 1e3
            keep = (eng.n_streams, eng.use_side)
            many = timed_ms(12)
            eng.n_streams, eng.use_side = 1, False
            run_device_steps(3)
            one = timed_ms(12)
            eng.n_streams, eng.use_side = keep
            eng._rr = 0
            line_extra["stream_overlap"] = {"one_stream_ms_per_step": round(one, 3), "engine_streams_ms_per_step": round(many, 3),
                                            "engine_streams": keep[0], "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES")}
        workload = ("BASELINE configs[1]: LSTM h=512, D-softmax* segs=(200,100,50), V=50k, beam=10, batch=256 sentences x 20 kana "
                    "per GPU; strings in -> n-best strings out (SURVEY 8d)"
                    if (args.fixture == "mid-vtable" and args.decoder == "static" and args.batch == 256 and args.length == 20
                        and args.beam == 10) else
                    "%s batch=%d length=%d beam=%d decoder=%s" % (args.fixture, args.batch, args.length, args.beam, args.decoder))
        scaling = "weak"
        cpu_sents, cpu_dec, cpu_root = sents, dec, root
        # ------------------------------------------------------------------ extra leg: BASELINE configs[4] at this N
        if not args.no_config5 and args.decoder == "static":
            _r5, _c5, alphabet5, dec5 = make_decoder("mid-tied", "static")
            dt5, chars5, n5 = run_config5(dec5, alphabet5, args.config5_passes)
            line_extra["config5"] = {
                "workload": "BASELINE configs[4]: tied softmax V=50k, beam=10, ONE set of %d sentences x %d kana sharded over %d GPU(s) "
                            "(jlm_amd.shard.decode_sharded), %d on this rank in chunks of 1 024; strings in -> n-best strings out" % (
                                CONFIG5_SENTENCES, args.length, world, n5),
                "value": round(chars5 * args.config5_passes / dt5, 1), "unit": "chars/s", "n_gpus": world, "scaling": "strong",
                "passes": args.config5_passes, "ms_per_pass": round(dt5 / args.config5_passes * 1e3, 2),
                "note": "`python bench.py --gpus N --config 5` makes this leg the headline line"}
            del dec5
            jconfig.set_root(root)

        # ---------------------------------------------- short legs: BASELINE configs[2] and configs[3] (one GPU, this process)
        if not args.no_legs and args.decoder == "static" and world == 1:
            def run_leg(fixture, kind, batch, beam, steps, kw, what, length=None, use=None, kernels=True):
                """one more workload, strings -> strings, settled like the headline; `use` = (root, alphabet, decoder): a decoder that
                exists already (the length legs run on the headline's)"""
                length = length or args.length
                if use is None:
                    root_, _cfg, alphabet_, dec_ = make_decoder(fixture, kind)
                else:
                    root_, alphabet_, dec_ = use
                    jconfig.set_root(root_)
                sents_ = synth.make_sentences(batch, length, seed=3131, alphabet=alphabet_)
                dec_.max_batch = batch
                n_settle_, prev_ = 0, None                       # untimed: calls of the timed size until two agree within 3 % (2-5)
                while n_settle_ < 5:
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    dec_.decode_batch(sents_ * steps, beam_width=beam, **kw)
                    torch.cuda.synchronize()
                    tq = time.perf_counter() - tq
                    n_settle_ += 1
                    if n_settle_ >= 2 and abs(tq - prev_) <= 0.03 * prev_:
                        break
                    prev_ = tq
                ta = time.perf_counter()
                out_ = dec_.decode_batch(sents_ * steps, beam_width=beam, **kw)
                torch.cuda.synchronize()
                dta = time.perf_counter() - ta
                assert len(out_) == batch * steps and all(len(r) > 0 for r in out_)
                del out_
                lat_ = BatchLattice(dec_._builder, sents_, beam)
                ekind_, ekw_ = ("dynamic", dict(dyn_lists=lat_.dynamic_vocab()[:4])) if kind == "dynamic" else ("static", {})
                # the same steps with the lattice resident (as the headline's device_resident_*)
                on_ = (dec_, lat_, ekind_, ekw_, batch, beam)
                run_device_steps(min(steps, 4), on=on_)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                run_device_steps(steps, on=on_)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - tb
                rf, gt = measure_kernels(dec_, lat_, ekind_, ekw_, 2, full_vocab=(kind == "static"), fixture=fixture, decoder_name=kind) if kernels else (None, None)
                leg = {"workload": what, "value": round(sum(len(x) for x in sents_) * steps / dta, 1), "unit": "chars/s", "n_gpus": 1,
                       "steps": steps, "untimed_steps": n_settle_ * steps, "ms_per_step": round(dta / steps * 1e3, 3),
                       "device_resident_ms_per_step": round(dtb / steps * 1e3, 3),
                       "timed": "strings -> strings, one decode_batch call of `steps` pipelined batches"}
                if rf:
                    leg["dominant_kernel"] = {k: rf.get(k) for k in ("kernel", "frac", "achieved", "peak", "unit", "frac_of_dense_f16",
                                                                     "avg_launch_ms", "note") if rf.get(k) is not None}
                    if rf.get("lse_form"):
                        leg["dominant_kernel"]["lse_form"] = rf["lse_form"]
                if gt:
                    leg["gate_gemm"] = {k: gt[k] for k in ("mfma_util_pct", "avg_launch_ms")}
                del dec_, lat_
                jconfig.set_root(root)
                return leg
            line_extra["config3"] = run_leg(
                "big-tied", "static", 1024, 20, 3, {},
                "BASELINE configs[2]: tied softmax V=100k (h=512, e=256), beam=20, batch=1024 sentences x %d kana" % args.length)
            # BASELINE configs[3] in a process of its own (this very script with --decoder dynamic): as the fourth decoder of this
            # process the incremental decoder -- the one whose host side is heaviest -- measured 1.71 ms per step where its own
            # process measures 1.33-1.36 (profiles/r05_l_switch_interval.txt); the leg reports what a service running it gets
            import subprocess
            what4 = ("BASELINE configs[3]: DynamicDecoder (decoder_dynamic.py incremental vocabulary selection), tied softmax V=50k, "
                     "beam=%d, batch=256 sentences x %d kana" % (args.beam, args.length))
            cmd4 = [sys.executable, os.path.abspath(__file__), "--fixture", "mid-tied", "--decoder", "dynamic", "--steps", "40", "--warmup", "3",
                    "--beam", str(args.beam), "--length", str(args.length), "--no-cpu-baseline", "--no-config5", "--no-legs"]
            try:
                pr = subprocess.run(cmd4, capture_output=True, text=True, timeout=600)
                d4 = json.loads(pr.stdout.strip().splitlines()[-1])
                line_extra["config4"] = {
                    "workload": what4, "value": d4["value"], "unit": "chars/s", "n_gpus": 1, "steps": d4["steps"],
                    "untimed_steps": d4.get("untimed_steps"), "ms_per_step": d4["ms_per_step"], "ms_per_step_repeats": d4.get("ms_per_step_repeats"),
                    "device_resident_ms_per_step": d4.get("device_resident_ms_per_step"),
                    "timed": "strings -> strings; `python bench.py --fixture mid-tied --decoder dynamic --steps 40 --warmup 3 --no-legs` in a process of its own",
                    "dominant_kernel": {k: (d4.get("roofline") or {}).get(k) for k in ("kernel", "avg_launch_ms", "note")},
                    "gate_gemm": {k: (d4.get("gate_gemm") or {}).get(k) for k in ("mfma_util_pct", "avg_launch_ms")}}
            except Exception as e:          # (a leg must not take the line with it)
                line_extra["config4"] = {"workload": what4, "error": "%s: %s" % (type(e).__name__, e)}

            # the stated sentence lengths ("also report L = 10, 40") on the headline's own model and decoder
            if args.fixture == "mid-vtable" and args.length == 20:
                for L_ in (10, 40):
                    line_extra["length%d" % L_] = run_leg(
                        args.fixture, "static", args.batch, args.beam, 20, {},
                        "BASELINE configs[1] at %d kana per sentence (batch=%d, beam=%d, D-softmax* V=50k)" % (L_, args.batch, args.beam),
                        length=L_, use=(root, alphabet, dec), kernels=False)
                # ... and what a TRAINED model gets: the same architecture with output embeddings x 20 (logits of +-20) and a unigram-like
                # bias (synth.shape_weights): its normaliser runs on split rows (DeviceModel._calibrate_mixed: the int8 cross terms
                # would cost 3e-6 rms per frame there), three f16 passes instead of one f16 + two int8
                line_extra["peaked20"] = run_leg(
                    "peaked20-vtable", "static", args.batch, args.beam, 20, {},
                    "BASELINE configs[1] on trained-model-like weights (fixture peaked20-vtable: logits of +-20, unigram-like bias), "
                    "batch=%d x %d kana, beam=%d" % (args.batch, args.length, args.beam))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    if not args.no_cpu_baseline and world == 1:
        from oracle import jlm_oracle as orc
        o = (orc.OracleDynamicDecoder if args.decoder == "dynamic" else orc.OracleDecoder)(cpu_root, 1)
        n = min(args.cpu_sentences, len(cpu_sents))
        # BLAS threads = the CPUs this job may use (the box shows 256 hardware threads behind a cgroup quota
        # of 16; more threads than that only get the process throttled)
        from jlm_amd import usable_cpus
        cores = usable_cpus()
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=cores)
        except ImportError:
            limit = None
        t2 = time.perf_counter()
        ref_out = [o.decode(s, beam_width=args.beam, **dkw) for s in cpu_sents[:n]]
        cdt = time.perf_counter() - t2
        if limit is not None:
            limit.restore_original_limits()
        gpu_out = cpu_dec.decode_batch(cpu_sents[:n], beam_width=args.beam, **dkw)
        same = sum(1 for a, b in zip(ref_out, gpu_out) if a[0][1] == b[0][1])
        cpu = {"value": round(sum(len(s) for s in cpu_sents[:n]) / cdt, 2), "unit": "chars/s", "cores": cores,
               "kind": "port",
               "sample": "%d of the step's %d sentences, sentence-at-a-time numpy oracle (oracle/jlm_oracle.py), "
                         "BLAS threads = usable CPUs (affinity capped by the cgroup quota; %d hardware threads visible); lstm %.1f%% / proj+softmax %.1f%% of its time; "
                         "1-best identical to the GPU path on %d/%d" % (
                             n, len(cpu_sents), os.cpu_count(), 100 * sum(o.perf_log_lstm) / cdt, 100 * sum(o.perf_log_softmax) / cdt, same, n)}

    m = cpu_dec.model.dev
    # north_star: throughput "as fraction of the gate-GEMM roofline" -- SURVEY 8(d): 2 (H + E) 4H flop per hypothesis row, beam rows per
    # decoded character (31.5 MFLOP per char at beam 10, E = 256), against the dense f16 matrix peak of N GPUs; and against the ceiling of
    # what the step kernel issues (three f16 passes over the state: the input side is a table row)
    _fpc = 2.0 * (m.H + m.E_in) * 4 * m.H * args.beam
    _issued = (SPLIT_PASSES if getattr(m, "split_lstm", False) else 1) * 2.0 * m.H * 4 * m.H * args.beam
    _peak = (F16_MFMA_PEAK_TFLOPS if getattr(m, "split_lstm", False) else F32_MFMA_PEAK_TFLOPS) * 1e12 * world
    line_extra["gate_roofline"] = {
        "flops_per_char": _fpc, "chars_per_s_at_peak": round(_peak / _fpc, 1), "frac": round(value / (_peak / _fpc), 5),
        "issued_flops_per_char": _issued, "chars_per_s_at_issued_ceiling": round(_peak / _issued, 1), "frac_of_issued_ceiling": round(value / (_peak / _issued), 5),
        "note": "value / (matrix peak of %d GPU(s) / gate-GEMM flop per decoded char): the whole decode priced as if it were the gate GEMM alone "
                "(SURVEY 8d); the step kernel's own utilisation is gate_gemm.mfma_util_pct" % world}
    from jlm_amd import ops as _ops
    assert type(_ops.backend()).__name__ == "HipOps", "bench.py measures the HIP path only (jlm_amd.ops.set_backend is a test hook)"
    line = {
        "metric": "decoded chars/sec at beam=%d, vocab=%dk (kana strings in -> n-best strings out, SURVEY 8d)" % (
            args.beam, cfg["vocab_size"] // 1000),
        "value": round(value, 1), "unit": "chars/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(steps_ms, 3),
        "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None,
        "dtype": ("f32 (matrix products as 3-pass split-f16 MFMA, f32 accumulate%s: f32-grade error, tests/test_gpu_kernels.py; "
                  "scores f64; parity bars: step logits <= 1e-4 relative, 1-best identical, path scores within 1e-6 per frame + 2e-6 of the "
                  "reference's -- tests/test_gpu_decode.py score_atol)" % (("; the vocabulary projection as f16 hi.hi + both cross terms in one block-scaled FP6 instruction per 32 k-values" if getattr(m, "mixed_fmt", None) == "mx6"
                                                                  else "; the vocabulary projection as f16 hi.hi + two int8 cross-term passes") if getattr(m, "mixed_idx", None) else "")
                  if getattr(m, "split_lstm", False) else "f32"),
        "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config, "fixture": args.fixture,
                   "sentences_per_gpu": args.batch if args.config == 2 else CONFIG5_SENTENCES // world,
                   "kana_per_sentence": args.length, "beam": args.beam, "decoder": args.decoder if args.config == 2 else "static",
                   "timed": "strings -> strings (lattice build, upload, device decode, n-best read-out), pipelined",
                   "parallelism": "sentence-sharded x%d, no collective" % world},
    }
    line["notes"] = ("north_star names wavefront shuffles for the back-pointer scan: the per-frame chain scan of the incremental decoder (beam_step_kernel<2>) "
                     "and the per-beam top-k (DPP arg-min rounds) are wave-level; the once-per-batch n-best trace is a wave per sentence since round 6 "
                     "(backtrace_wave_kernel: the sentence's back-pointer table in registers, paths walked by ds_bpermute shuffles; 6.9 vs 12.1 us)")
    line.update(line_extra)
    line.update({"roofline": roofline, "gate_gemm": gate_obj, "cpu_baseline": cpu})
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
