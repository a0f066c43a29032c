// jlm_torch_ops.cpp -- the PyTorch-ROCm custom-op boundary of the decode path.
//
// A thin TORCH_LIBRARY shim over the C-ABI launchers of libjlm_hip.so (include/jlm_hip.h): torch owns every device buffer,
// the ops take torch.Tensor arguments (never raw pointers from Python), launch on the CURRENT torch HIP stream of the
// tensors' device, and turn a non-zero return into a c10::Error (TORCH_CHECK) carrying hipGetErrorString.  What it
// replaces on the reference's side are the call sites decoder/decoder.py:202-218 -> decoder/model.py:195-198
// (Decoder._batch_predict -> LSTM_Model.predict_with_context) and the frame loop around them (decoder.py:220-241).
//
//   torch.classes.jlm.Model   the model's weight panels as jlm_decode_model (keeps the tensors alive)
//   torch.classes.jlm.Plan    the buffers of one decode shape as jlm_lattice / jlm_beam_state / jlm_decode_plan,
//                             plus the timing events of a timed decode
//   torch.ops.jlm.decode_frames(Model, Plan, n_frames, vs_max, di_max, dd_max, use_side, timed, lse_cu_share_pct)
//                             the whole frame loop of a batch: ONE op (jlm_decode_frames)
//   torch.ops.jlm.decode_batch(Model, Plan, staging block, lattice block, read-back buffers, ...)
//                             round 5: upload + frame loop + read-back of one batch, ONE op (what DecodeEngine submits)
//   torch.ops.jlm.frame_times(Plan) -> Tensor [n_frames, 5] milliseconds of the last timed decode (after it finished)
//   torch.ops.jlm.lstm_step / gemm_nt / softmax_rows      LSTM_Model.predict / project (numpy-facing API)
//   torch.ops.jlm.pack_split_f16 / pack_split_f16_col / dequant_u8     weight preparation at load
//
// Built by __graft_entry__.build():  g++ -shared ... -> jlm_amd/_torch_ops.so, loaded with torch.ops.load_library.
#include <torch/library.h>
#include <torch/custom_class.h>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>
#include <hip/hip_runtime_api.h>

#include <array>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/jlm_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;
using TDict = c10::Dict<std::string, Tensor>;
using IDict = c10::Dict<std::string, int64_t>;
using FDict = c10::Dict<std::string, double>;

void jlm_check(int rc, const char *what) {
    if (rc == 0) return;
    if (rc > 0)
        TORCH_CHECK(false, what, " failed: ", hipGetErrorString(static_cast<hipError_t>(rc)), " (hipError_t ", rc, ")");
    TORCH_CHECK(false, what, " rejected its arguments (code ", rc,
                ": -1 = shape / alignment the kernels cannot handle, -2 = outside the shapes this entry point covers, "
                "-3 = the kernel's LDS size could not be set; include/jlm_hip.h)");
}

void on_gpu(const Tensor &t, const char *name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), "jlm: `", name, "` must be a tensor on the GPU (there is no CPU path)");
    TORCH_CHECK(t.is_contiguous(), "jlm: `", name, "` must be contiguous");
}

template <class T = void> T *ptr(const Tensor &t, const char *name) {
    on_gpu(t, name);
    return reinterpret_cast<T *>(t.data_ptr());
}
template <class T = void> T *optr(const OptTensor &t, const char *name) {
    if (!t.has_value() || !t->defined()) return nullptr;
    return ptr<T>(*t, name);
}

// the current torch stream of the device the tensor lives on (ops never switch devices themselves: the caller's
// torch.cuda.device context / set_device is what the HIP runtime launches on)
hipStream_t stream_of(const Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

struct Segments {
    std::vector<jlm_segment> v;
    std::vector<Tensor> keep;
    Segments() = default;
    // meta: (v_start, v_end, k, t_off, ldb) per segment
    Segments(const std::vector<Tensor> &B, const std::vector<int64_t> &meta) : keep(B) {
        TORCH_CHECK(meta.size() == 5 * B.size() && B.size() <= JLM_MAX_SEGMENTS, "jlm: malformed segment table");
        for (size_t i = 0; i < B.size(); ++i) {
            jlm_segment s;
            s.v_start = (int)meta[5 * i]; s.v_end = (int)meta[5 * i + 1]; s.k = (int)meta[5 * i + 2];
            s.t_off = (int)meta[5 * i + 3]; s.ldb = (int)meta[5 * i + 4];
            s.B = ptr<const float>(B[i], "segment block");
            v.push_back(s);
        }
    }
};

const Tensor *find(const TDict &d, const char *key) {
    auto it = d.find(key);
    return it == d.end() ? nullptr : &it->value();
}
int64_t geti(const IDict &d, const char *key, int64_t dflt = 0) {
    auto it = d.find(key);
    return it == d.end() ? dflt : it->value();
}
double getf(const FDict &d, const char *key, double dflt = 0.0) {
    auto it = d.find(key);
    return it == d.end() ? dflt : it->value();
}
template <class T = void> T *tptr(const TDict &d, const char *key) {
    const Tensor *t = find(d, key);
    return t ? ptr<T>(*t, key) : nullptr;
}

// ------------------------------------------------------------------------------------------------- Model
struct JlmModel : torch::CustomClassHolder {
    TDict tensors;
    Segments segs, split, mixed_some;
    std::vector<float> t_scale, descale;
    std::vector<int> bias_col;
    std::vector<jlm_segment> mixed;                 // n_segs entries, B == NULL where the segment has no mixed rows
    std::vector<float> mx_t_scale, mx_descale, mx_s8;
    std::vector<int> mx_head_split;                 // n_segs entries (ABI 10)
    jlm_decode_model m{};
    // mixed_*: the segments of the full-vocabulary normaliser that also exist as mixed rows (ABI 7): their indices, blocks,
    // (v_start, v_end, k, t_off, ldb) and the three scales, one entry per such segment
    JlmModel(TDict t, IDict i, FDict f, std::vector<Tensor> seg_B, std::vector<int64_t> seg_meta, std::vector<Tensor> split_B,
             std::vector<int64_t> split_meta, std::vector<double> split_t_scale, std::vector<double> split_descale,
             std::vector<int64_t> split_bias_col, std::vector<int64_t> mixed_idx, std::vector<Tensor> mixed_B,
             std::vector<int64_t> mixed_meta, std::vector<double> mixed_t_scale, std::vector<double> mixed_descale,
             std::vector<double> mixed_s8, std::vector<int64_t> mixed_head_split)
        : tensors(std::move(t)), segs(seg_B, seg_meta), split(split_B, split_meta), mixed_some(mixed_B, mixed_meta) {
        for (double x : split_t_scale) t_scale.push_back((float)x);
        for (double x : split_descale) descale.push_back((float)x);
        for (int64_t x : split_bias_col) bias_col.push_back((int)x);
        TORCH_CHECK(t_scale.size() == split.v.size() && descale.size() == split.v.size() && bias_col.size() == split.v.size(),
                    "jlm.Model: one scale / descale / bias column per split segment");
        m.segs = segs.v.data(); m.n_segs = (int)segs.v.size();
        m.b2 = tptr<const float>(tensors, "b2");
        m.H = (int)geti(i, "H"); m.ldt = (int)geti(i, "ldt");
        m.untied = (int)geti(i, "untied"); m.self_norm = (int)geti(i, "self_norm"); m.split_lstm = (int)geti(i, "split_lstm");
        m.emb = tptr<const float>(tensors, "emb"); m.ld_emb = (int)geti(i, "ld_emb");
        m.wt = tptr<const float>(tensors, "wt"); m.gate_bias = tptr<const float>(tensors, "gate_bias");
        m.kpad = (int)geti(i, "kpad"); m.E = (int)geti(i, "E");
        m.gate_descale = (float)getf(f, "gate_descale"); m.h_scale = (float)getf(f, "h_scale");
        m.wt8 = tptr<const void>(tensors, "wt8"); m.xgate8 = tptr<const float>(tensors, "xgate8");
        m.untied_split = tptr<const void>(tensors, "untied_split"); m.untied_descale = (float)getf(f, "untied_descale");
        m.lse_fixed_ref = (int)geti(i, "lse_fixed_ref");
        m.pmt = tptr<const float>(tensors, "pmt"); m.pmt_split = tptr<const void>(tensors, "pmt_split");
        m.n_t = (int)geti(i, "n_t"); m.t_descale = (float)getf(f, "t_descale");
        if (!split.v.empty()) {
            m.split_segs = split.v.data(); m.split_t_scale = t_scale.data(); m.split_descale = descale.data();
            m.split_bias_col = bias_col.data();
        }
        if (!mixed_idx.empty()) {
            const size_t n = mixed_idx.size();
            // (an untied model has no rows-stationary split table: its one segment, k = H, is checked against the f32 segment)
            TORCH_CHECK((!split.v.empty() || m.untied) && mixed_some.v.size() == n && mixed_t_scale.size() == n && mixed_descale.size() == n &&
                            mixed_s8.size() == n && (mixed_head_split.empty() || mixed_head_split.size() == n),
                        "jlm.Model: mixed segments need the split table and one block / scale / descale / s8 (/ head_split) each");
            jlm_segment none{};
            mixed.assign(segs.v.size(), none);
            mx_t_scale.assign(segs.v.size(), 0.0f); mx_descale.assign(segs.v.size(), 0.0f); mx_s8.assign(segs.v.size(), 0.0f);
            mx_head_split.assign(segs.v.size(), 0);
            for (size_t j = 0; j < n; ++j) {
                const int64_t si = mixed_idx[j];
                TORCH_CHECK(si >= 0 && si < (int64_t)segs.v.size() && !mixed[si].B, "jlm.Model: bad mixed segment index");
                const jlm_segment &a = mixed_some.v[j], &b = split.v.empty() ? segs.v[si] : split.v[si];
                TORCH_CHECK(a.v_start == b.v_start && a.v_end == b.v_end && a.k == b.k && a.t_off == b.t_off,
                            "jlm.Model: a mixed segment must describe the same words and T columns as its split form");
                TORCH_CHECK(mixed_some.keep[j].numel() >= (int64_t)(a.v_end - a.v_start) * a.ldb, "jlm.Model: mixed block too small");
                mixed[si] = a;
                mx_t_scale[si] = (float)mixed_t_scale[j]; mx_descale[si] = (float)mixed_descale[j]; mx_s8[si] = (float)mixed_s8[j];
                if (!mixed_head_split.empty()) {
                    const int64_t c = mixed_head_split[j];
                    TORCH_CHECK(c >= 0 && c % 128 == 0 && c < a.v_end - a.v_start && (c == 0 || !split.v.empty()),
                                "jlm.Model: head_split is a multiple of 128 below the segment's size, and needs the split rows");
                    mx_head_split[si] = (int)c;
                }
            }
            m.mixed_head_split = mx_head_split.data();
            m.mixed_segs = mixed.data(); m.mixed_t_scale = mx_t_scale.data(); m.mixed_descale = mx_descale.data();
            m.mixed_s8 = mx_s8.data();
            m.mixed_bias2 = tptr<const float>(tensors, "b2_log2");
        }
        TORCH_CHECK(m.b2 && m.n_segs >= 1 && m.H > 0, "jlm.Model: b2, the segments and H are required");
        TORCH_CHECK(!m.split_lstm || (m.wt8 && m.xgate8 && (m.pmt_split || m.untied)), "jlm.Model: split_lstm needs wt8, xgate8, pmt_split");
    }
};

// ------------------------------------------------------------------------------------------------- Plan
struct JlmPlan : torch::CustomClassHolder {
    TDict tensors;
    jlm_lattice lat{};
    jlm_beam_state st{};
    jlm_decode_plan p{};
    int frames_cap = 0;
    std::vector<hipEvent_t> events;         // JLM_EVENTS_PER_FRAME per frame, created on first timed decode
    int timed_frames = 0;                   // frames of the last timed decode (0: the last decode was not timed)
    int device = -1;

    JlmPlan(TDict t, IDict i) : tensors(std::move(t)) {
        const Tensor *ints = find(tensors, "ints");
        TORCH_CHECK(ints && ints->scalar_type() == at::kInt, "jlm.Plan: `ints` (int32 staging block) is required");
        device = ints->device().index();
        const int *base = ptr<const int>(*ints, "ints");
        auto at_off = [&](const char *key) -> const int * {
            auto it = i.find(key);
            TORCH_CHECK(it != i.end(), "jlm.Plan: missing offset ", key);
            TORCH_CHECK(it->value() >= 0 && it->value() < ints->numel(), "jlm.Plan: offset ", key, " outside `ints`");
            return base + it->value();
        };
        lat.n_sent = (int)geti(i, "n_sent"); lat.beam = (int)geti(i, "beam"); lat.n_frames = 0;
        frames_cap = (int)geti(i, "frames");
        lat.sent_len = at_off("off_sent_len"); lat.end_off = at_off("off_end_off");
        lat.node_start = at_off("off_node_start"); lat.node_word = at_off("off_node_word");
        st.score = tptr<double>(tensors, "score"); st.lse = tptr<double>(tensors, "lse"); st.ysum = tptr<double>(tensors, "ysum");
        st.bp = tptr<int>(tensors, "bp"); st.node = tptr<int>(tensors, "node"); st.word = tptr<int>(tensors, "word");
        st.cnt = tptr<int>(tensors, "cnt"); st.live = tptr<int>(tensors, "live"); st.n_live = tptr<int>(tensors, "n_live");
        st.edge = tptr<const float>(tensors, "edge"); st.live_base = tptr<int>(tensors, "live_base");
        st.lse_part = nullptr; st.ld_part = 0; st.n_parts = 0; st.flags = nullptr;
        p.kind = (int)geti(i, "kind"); p.max_cands = (int)geti(i, "max_cands");
        p.h = tptr<void>(tensors, "h"); p.c = tptr<float>(tensors, "c"); p.T = tptr<float>(tensors, "T");
        p.g0 = at_off("off_g0"); p.cidx = at_off("off_cidx"); p.sidx = at_off("off_sidx");
        p.sg_word = at_off("off_sg_word"); p.sg_off = at_off("off_sg_off"); p.sg_node = at_off("off_sg_node");
        p.edge = tptr<float>(tensors, "edge");
        p.vs_words = at_off("off_vs_words"); p.vs_off = at_off("off_vs_off");
        p.di_words = at_off("off_di_words"); p.di_off = at_off("off_di_off"); p.di_idx = at_off("off_sidx2");
        p.dd_words = at_off("off_dd_words"); p.dd_off = at_off("off_dd_off");
        // reference-compatibility lists of the incremental decoder on segmented models: present only in such plans
        p.di_wwords = i.find("off_di_wwords") != i.end() ? at_off("off_di_wwords") : nullptr;
        p.sg_wword = i.find("off_sg_wword") != i.end() ? at_off("off_sg_wword") : nullptr;
        p.run_max = tptr<float>(tensors, "run_max"); p.run_sum = tptr<double>(tensors, "run_sum");
        p.part = tptr<float>(tensors, "part"); p.max_parts = (int)geti(i, "max_parts");
        p.Tm = tptr<void>(tensors, "Tm"); p.ld_tm = (int)geti(i, "ld_tm");
        TORCH_CHECK(!p.Tm || (p.ld_tm > 0 && find(tensors, "Tm")->numel() >= (((int64_t)lat.n_sent * lat.beam + 31) / 32 * 32) * p.ld_tm),
                    "jlm.Plan: Tm must hold n_sent * beam rows (rounded up to whole 32-row blocks) of ld_tm");
        p.out_nodes = tptr<int>(tensors, "out_nodes"); p.out_len = tptr<int>(tensors, "out_len");
        p.out_score = tptr<double>(tensors, "out_score"); p.stride = (int)geti(i, "stride");
        // ABI 11: one spare element behind the trace lengths = the batch's flag word (jlm_beam_state.flags): it travels back with them
        if (p.out_len && find(tensors, "out_len")->numel() > (int64_t)lat.n_sent * lat.beam) st.flags = p.out_len + (size_t)lat.n_sent * lat.beam;
        TORCH_CHECK(st.score && st.lse && st.bp && st.node && st.word && st.cnt && st.live && st.n_live && st.live_base && p.h && p.c &&
                        p.T && p.edge && p.out_nodes && p.out_len && p.out_score && lat.n_sent > 0 && lat.beam > 0 && frames_cap > 0,
                    "jlm.Plan: a required buffer is missing");
    }
    ~JlmPlan() override {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};

// side stream of each launch stream (edge logits beside the normaliser): ONE per launch stream for the whole process -- ROCm
// maps streams onto GPU_MAX_HW_QUEUES hardware queues and streams that share a queue serialise (jlm_amd/__init__.py)
std::map<std::pair<int, hipStream_t>, c10::hip::HIPStream> g_side;
// torch releases the interpreter lock around custom ops: two Python threads may be inside decode_frames at once (two
// decoders).  The side-stream table, a plan's event table and the launchers' one-time kernel
// attributes (function-local statics in libjlm_hip.so) are all touched in here: one lock around the enqueue.  It is held
// for the ~0.5 ms the launches take; the GPU work itself is asynchronous.
std::mutex g_enqueue_mutex;

int64_t decode_frames(const c10::intrusive_ptr<JlmModel> &model, const c10::intrusive_ptr<JlmPlan> &plan, int64_t n_frames,
                      int64_t vs_max, int64_t di_max, int64_t dd_max, bool use_side, bool timed, int64_t lse_cu_share_pct) {
    JlmPlan &pl = *plan;
    TORCH_CHECK(n_frames >= 1 && n_frames <= pl.frames_cap, "jlm.decode_frames: ", n_frames, " frames, the plan holds ", pl.frames_cap);
    const c10::hip::HIPGuard device_guard(pl.device);          // the plan's device, whatever the caller's current one is
    const std::lock_guard<std::mutex> lock(g_enqueue_mutex);
    c10::hip::HIPStream main = c10::hip::getCurrentHIPStream(pl.device);
    pl.lat.n_frames = (int)n_frames;
    pl.p.vs_max = (int)vs_max; pl.p.di_max = (int)di_max; pl.p.dd_max = (int)dd_max;
    pl.p.lse_cu_share_pct = (int)lse_cu_share_pct;
    void *side_s = nullptr;
    if (use_side && !timed) {
        const auto key = std::make_pair(pl.device, main.stream());
        auto it = g_side.find(key);
        if (it == g_side.end()) it = g_side.emplace(key, c10::hip::getStreamFromPool(false, pl.device)).first;
        side_s = it->second.stream();
    }
    void *const *ev = nullptr;
    pl.timed_frames = 0;
    if (timed) {
        const size_t need = (size_t)n_frames * JLM_EVENTS_PER_FRAME;
        while (pl.events.size() < need) {
            hipEvent_t e;
            jlm_check((int)hipEventCreate(&e), "hipEventCreate");
            pl.events.push_back(e);
        }
        ev = reinterpret_cast<void *const *>(pl.events.data());
        pl.timed_frames = (int)n_frames;
    }
    // (Rounds 2-4 could replay the launch sequence as a captured hipGraph, JLM_GRAPH=1: correct and slower -- 3.52-3.58 vs 2.58-2.71 ms
    //  per 256-sentence chunk, hipGraphLaunch of the 130-node two-branch graph cost more than the launches -- removed in round 5.)
    const int rc = jlm_decode_frames(&model->m, &pl.p, &pl.lat, &pl.st, main.stream(), side_s, ev);
    if (rc == -2) return -2;
    jlm_check(rc, "jlm_decode_frames");
    return 0;
}

// Round 5: one batch, one op -- everything DecodeEngine._enqueue did between the staging block and the `done` event as a
// dozen torch calls (0.3-0.6 ms of interpreter time per 256-sentence chunk, tools/probes/host_profile2.py): the upload of the
// batch's lattice (head of the plan's page-locked staging block; the four node arrays straight from the lattice's own
// page-locked block), the counters' reset, the frame loop, and the read-back of the n-best traces into page-locked host
// buffers.  All asynchronous on the current stream; torch has released the interpreter lock around the whole call, so a
// second Python thread (the collector building the previous batch's strings) runs beside it.
//   host_ints [>= head_end] int32, page-locked: the staging block; blk (optional) int32, page-locked: the lattice's block, of
//   which blk_n[i] ints at blk_src[i] go to dev_ints + blk_dst[i]; without blk the whole staging block is copied.
//   h_nodes / h_len / h_score (/ h_nlive with a timed decode): page-locked destinations of out_nodes / out_len / out_score / n_live.
int64_t decode_batch(const c10::intrusive_ptr<JlmModel> &model, const c10::intrusive_ptr<JlmPlan> &plan, const Tensor &host_ints,
                     int64_t head_end, const OptTensor &blk, std::vector<int64_t> blk_src, std::vector<int64_t> blk_dst,
                     std::vector<int64_t> blk_n, const Tensor &h_nodes, const Tensor &h_len, const Tensor &h_score, const OptTensor &h_nlive,
                     int64_t n_frames, int64_t vs_max, int64_t di_max, int64_t dd_max, bool use_side, bool timed, int64_t lse_cu_share_pct) {
    JlmPlan &pl = *plan;
    const Tensor *dev_ints = find(pl.tensors, "ints");
    const Tensor *cnt = find(pl.tensors, "cnt"), *n_live = find(pl.tensors, "n_live");
    const Tensor *o_nodes = find(pl.tensors, "out_nodes"), *o_len = find(pl.tensors, "out_len"), *o_score = find(pl.tensors, "out_score");
    auto host_ok = [](const Tensor &t, at::ScalarType ty) { return t.defined() && !t.is_cuda() && t.is_contiguous() && t.scalar_type() == ty; };
    TORCH_CHECK(host_ok(host_ints, at::kInt) && host_ints.numel() <= dev_ints->numel() && head_end >= 0 && head_end <= host_ints.numel(),
                "jlm.decode_batch: the staging block must be a contiguous int32 host tensor no larger than the plan's");
    TORCH_CHECK(host_ok(h_nodes, at::kInt) && h_nodes.numel() >= o_nodes->numel() && host_ok(h_len, at::kInt) && h_len.numel() >= o_len->numel() &&
                    host_ok(h_score, at::kDouble) && h_score.numel() >= o_score->numel(),
                "jlm.decode_batch: read-back buffers");
    const bool has_blk = blk.has_value() && blk->defined();
    TORCH_CHECK(!has_blk || (host_ok(*blk, at::kInt) && blk_src.size() == blk_dst.size() && blk_src.size() == blk_n.size()), "jlm.decode_batch: lattice block");
    const c10::hip::HIPGuard device_guard(pl.device);
    hipStream_t st = c10::hip::getCurrentHIPStream(pl.device).stream();
    int *d = reinterpret_cast<int *>(dev_ints->data_ptr());
    const int *h = reinterpret_cast<const int *>(host_ints.data_ptr());
    if (!has_blk) {
        jlm_check((int)hipMemcpyAsync(d, h, (size_t)host_ints.numel() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync (staging block)");
    } else {
        if (head_end) jlm_check((int)hipMemcpyAsync(d, h, (size_t)head_end * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync (staging head)");
        const int *b = reinterpret_cast<const int *>(blk->data_ptr());
        for (size_t i = 0; i < blk_n.size(); ++i) {
            if (blk_n[i] <= 0) continue;
            TORCH_CHECK(blk_src[i] >= 0 && blk_src[i] + blk_n[i] <= blk->numel() && blk_dst[i] >= 0 && blk_dst[i] + blk_n[i] <= dev_ints->numel(),
                        "jlm.decode_batch: a lattice array outside its block or the plan");
            jlm_check((int)hipMemcpyAsync(d + blk_dst[i], b + blk_src[i], (size_t)blk_n[i] * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync (lattice array)");
        }
    }
    jlm_check((int)hipMemsetAsync(cnt->data_ptr(), 0, (size_t)cnt->numel() * 4, st), "hipMemsetAsync (cnt)");
    jlm_check((int)hipMemsetAsync(n_live->data_ptr(), 0, (size_t)n_live->numel() * 4, st), "hipMemsetAsync (n_live)");
    if (pl.st.flags) jlm_check((int)hipMemsetAsync(pl.st.flags, 0, 4, st), "hipMemsetAsync (flags)");
    const int64_t rc = decode_frames(model, plan, n_frames, vs_max, di_max, dd_max, use_side, timed, lse_cu_share_pct);
    if (rc != 0) return rc;
    jlm_check((int)hipMemcpyAsync(h_nodes.data_ptr(), o_nodes->data_ptr(), (size_t)o_nodes->numel() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (traces)");
    jlm_check((int)hipMemcpyAsync(h_len.data_ptr(), o_len->data_ptr(), (size_t)o_len->numel() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (trace lengths)");
    jlm_check((int)hipMemcpyAsync(h_score.data_ptr(), o_score->data_ptr(), (size_t)o_score->numel() * 8, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (scores)");
    if (h_nlive.has_value() && h_nlive->defined()) {
        TORCH_CHECK(host_ok(*h_nlive, at::kInt) && h_nlive->numel() >= n_live->numel(), "jlm.decode_batch: live-row read-back buffer");
        jlm_check((int)hipMemcpyAsync(h_nlive->data_ptr(), n_live->data_ptr(), (size_t)n_live->numel() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (live rows)");
    }
    return 0;
}

// [n_frames, 5] milliseconds of the plan's last TIMED decode, which must have finished (the caller synchronised on it):
// vocab fix, lattice path fix (beam step), LSTM step, T projection + edge logits, normaliser (include/jlm_hip.h, events)
Tensor frame_times(const c10::intrusive_ptr<JlmPlan> &plan) {
    JlmPlan &pl = *plan;
    const c10::hip::HIPGuard device_guard(pl.device);
    const std::lock_guard<std::mutex> lock(g_enqueue_mutex);
    const int F = pl.timed_frames;
    Tensor out = at::zeros({F, 5}, at::kDouble);
    auto a = out.accessor<double, 2>();
    for (int f = 0; f < F; ++f) {
        const int last = (f == F - 1) ? 2 : 5;          // the last frame is not stepped
        for (int i = 0; i < last; ++i) {
            float ms = 0.0f;
            jlm_check((int)hipEventElapsedTime(&ms, pl.events[(size_t)f * JLM_EVENTS_PER_FRAME + i],
                                               pl.events[(size_t)f * JLM_EVENTS_PER_FRAME + i + 1]), "hipEventElapsedTime");
            a[f][i] = ms;
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------ LSTM_Model.predict / project
void lstm_step(const Tensor &h_in, const Tensor &c_in, int64_t ld_state, const Tensor &h_out, const Tensor &c_out, const OptTensor &rows,
               const Tensor &prev, const Tensor &word, const Tensor &emb, int64_t ld_emb, const Tensor &wt, const Tensor &bias,
               int64_t kpad, int64_t H, int64_t E, int64_t n_rows_max, const OptTensor &n_dev) {
    jlm_check(jlm_lstm_step(ptr<const float>(h_in, "h_in"), ptr<const float>(c_in, "c_in"), (int)ld_state, ptr<float>(h_out, "h_out"),
                            ptr<float>(c_out, "c_out"), optr<const int>(rows, "rows"), ptr<const int>(prev, "prev"),
                            ptr<const int>(word, "word"), ptr<const float>(emb, "emb"), (int)ld_emb, ptr<const float>(wt, "wt"),
                            ptr<const float>(bias, "bias"), (int)kpad, (int)H, (int)E, (int)n_rows_max, optr<const int>(n_dev, "n_dev"),
                            stream_of(h_in)),
              "jlm_lstm_step");
}

// C[c_rows[m], n] = sum_k A[a_rows[m], k] * B[b_rows[n], k] + bias[n]; a_off / c_off / bias_off: element offsets of the
// views inside their storage (a column range of T, of the logits, of b2)
void gemm_nt(const Tensor &A, int64_t a_off, int64_t lda, const OptTensor &a_rows, const Tensor &B, int64_t ldb, const OptTensor &b_rows,
             const Tensor &C, int64_t c_off, int64_t ldc, const OptTensor &c_rows, const OptTensor &bias, int64_t bias_off, int64_t M,
             int64_t N, int64_t K, const OptTensor &m_dev) {
    TORCH_CHECK(a_off >= 0 && a_off < A.numel() && c_off >= 0 && c_off < C.numel(), "jlm.gemm_nt: offset outside the tensor");
    const float *bp = optr<const float>(bias, "bias");
    jlm_check(jlm_gemm_nt(ptr<const float>(A, "A") + a_off, (int)lda, optr<const int>(a_rows, "a_rows"), ptr<const float>(B, "B"), (int)ldb,
                          optr<const int>(b_rows, "b_rows"), ptr<float>(C, "C") + c_off, (int)ldc, optr<const int>(c_rows, "c_rows"),
                          bp ? bp + bias_off : nullptr, (int)M, (int)N, (int)K, optr<const int>(m_dev, "m_dev"), stream_of(A)),
              "jlm_gemm_nt");
}

void softmax_rows(const Tensor &y, const Tensor &pred, int64_t ld, int64_t n_rows, int64_t n_cols, bool self_norm) {
    jlm_check(jlm_softmax_rows(ptr<const float>(y, "y"), ptr<float>(pred, "pred"), (int)ld, (int)n_rows, (int)n_cols, self_norm ? 1 : 0,
                               stream_of(y)),
              "jlm_softmax_rows");
}

// ------------------------------------------------------------------------------------ weight preparation
void pack_split_f16(const Tensor &src, int64_t src_off, int64_t rows, int64_t k, int64_t ld, double scale, const Tensor &dst,
                    int64_t dst_off, int64_t ld_dst) {
    TORCH_CHECK(src_off >= 0 && src_off < src.numel() && dst_off >= 0 && dst_off < dst.numel(), "jlm.pack_split_f16: offset outside the tensor");
    jlm_check(jlm_pack_split_f16(ptr<const float>(src, "src") + src_off, (int)rows, (int)k, (int)ld, (float)scale,
                                 ptr<float>(dst, "dst") + dst_off, (int)ld_dst, stream_of(src)),
              "jlm_pack_split_f16");
}

void pack_split_f16_col(const Tensor &v, int64_t v_off, int64_t rows, double scale, const Tensor &dst, int64_t ld_dst, int64_t col) {
    TORCH_CHECK(v_off >= 0 && v_off + rows <= v.numel(), "jlm.pack_split_f16_col: range outside the vector");
    jlm_check(jlm_pack_split_f16_col(ptr<const float>(v, "v") + v_off, (int)rows, (float)scale, ptr<float>(dst, "dst"), (int)ld_dst, (int)col,
                                     stream_of(v)),
              "jlm_pack_split_f16_col");
}

// mixed rows of a vocabulary block (include/jlm_hip.h ABI 7): src [rows, k] f32 + the words' biases -> dst [rows, ld_dst]
void pack_mixed(const Tensor &src, int64_t src_off, int64_t rows, int64_t k, int64_t ld, const Tensor &bias, int64_t bias_off, double scale,
                double bias_scale, double s8, const Tensor &dst, int64_t ld_dst) {
    TORCH_CHECK(src_off >= 0 && rows >= 0 && (rows == 0 || src_off + (rows - 1) * ld + k <= src.numel()), "jlm.pack_mixed: source range");
    TORCH_CHECK(bias_off >= 0 && bias_off + rows <= bias.numel(), "jlm.pack_mixed: bias range");
    TORCH_CHECK(ld_dst % 32 == 0 && (ld_dst == (k + 2 + 31) / 32 * 32 || ld_dst == (k + 31) / 32 * 32) && rows * ld_dst <= dst.numel(),
                "jlm.pack_mixed: destination shape");
    jlm_check(jlm_pack_mixed(ptr<const float>(src, "src") + src_off, (int)rows, (int)k, (int)ld, ptr<const float>(bias, "bias") + bias_off,
                             (float)scale, (float)bias_scale, (float)s8, ptr<void>(dst, "dst"), (int)ld_dst, stream_of(src)),
              "jlm_pack_mixed");
}

// dst[r][c] = codebook[code[r][c]]: the k-means (code, codebook) form of a weight tensor expanded on the device
void dequant_u8(const Tensor &code, int64_t rows, int64_t k, int64_t ld_code, const Tensor &codebook, const Tensor &dst, int64_t ld_dst) {
    TORCH_CHECK(code.scalar_type() == at::kByte && codebook.scalar_type() == at::kFloat && dst.scalar_type() == at::kFloat,
                "jlm.dequant_u8: uint8 codes, float32 codebook and destination");
    TORCH_CHECK(rows * ld_code <= code.numel() && (rows == 0 || (rows - 1) * ld_dst + k <= dst.numel()), "jlm.dequant_u8: shape outside the tensors");
    jlm_check(jlm_dequant_u8(ptr<const uint8_t>(code, "code"), (int)rows, (int)k, (int)ld_code, ptr<const float>(codebook, "codebook"),
                             (int)codebook.numel(), ptr<float>(dst, "dst"), (int)ld_dst, stream_of(code)),
              "jlm_dequant_u8");
}

// the normaliser slices of a few probe rows in one of its forms (jlm_lse_probe, include/jlm_hip.h ABI 8): -> number of slices, -2
// when the model has no such form
int64_t lse_probe(const c10::intrusive_ptr<JlmModel> &model, const Tensor &rowlist, const Tensor &prev, const Tensor &word, int64_t steps,
                  int64_t rows, const Tensor &h, const Tensor &c, const Tensor &T, const OptTensor &Tm, int64_t ld_tm, int64_t form,
                  const Tensor &part, int64_t max_parts) {
    const int64_t G = (steps + 1) * rows;
    TORCH_CHECK(steps >= 1 && rows >= 1 && rowlist.numel() >= G && prev.numel() >= G && word.numel() >= G, "jlm.lse_probe: index arrays");
    TORCH_CHECK(rowlist.scalar_type() == at::kInt && prev.scalar_type() == at::kInt && word.scalar_type() == at::kInt, "jlm.lse_probe: int32 indices");
    TORCH_CHECK(h.numel() >= G * model->m.H && c.numel() >= G * model->m.H && T.numel() >= G * model->m.ldt, "jlm.lse_probe: state buffers");
    TORCH_CHECK(part.numel() >= max_parts * rows * 2 && max_parts >= 1, "jlm.lse_probe: slice buffer");
    TORCH_CHECK(!Tm.has_value() || !Tm->defined() || Tm->numel() >= ((rows + 31) / 32 * 32) * ld_tm, "jlm.lse_probe: packed-row buffer (whole 32-row blocks)");
    const c10::hip::HIPGuard device_guard(h.device().index());
    const std::lock_guard<std::mutex> lock(g_enqueue_mutex);
    const int rc = jlm_lse_probe(&model->m, ptr<const int>(rowlist, "rowlist"), ptr<const int>(prev, "prev"), ptr<const int>(word, "word"),
                                 (int)steps, (int)rows, ptr<void>(h, "h"), ptr<float>(c, "c"), ptr<float>(T, "T"), optr<void>(Tm, "Tm"),
                                 (int)ld_tm, (int)form, ptr<float>(part, "part"), (int)max_parts, stream_of(h));
    if (rc == -2 || rc >= 1) return rc;
    jlm_check(rc == 0 ? -1 : rc, "jlm_lse_probe");
    return rc;
}

int64_t abi_version() { return jlm_abi_version(); }
int64_t beam_step_max_cands(int64_t beam, int64_t n_frames, int64_t mode) { return jlm_beam_step_max_cands((int)beam, (int)n_frames, (int)mode); }

}  // namespace

TORCH_LIBRARY(jlm, m) {
    m.class_<JlmModel>("Model").def(
        torch::init<TDict, IDict, FDict, std::vector<Tensor>, std::vector<int64_t>, std::vector<Tensor>, std::vector<int64_t>,
                    std::vector<double>, std::vector<double>, std::vector<int64_t>, std::vector<int64_t>, std::vector<Tensor>,
                    std::vector<int64_t>, std::vector<double>, std::vector<double>, std::vector<double>, std::vector<int64_t>>());
    m.class_<JlmPlan>("Plan").def(torch::init<TDict, IDict>());
    m.def("decode_frames(__torch__.torch.classes.jlm.Model model, __torch__.torch.classes.jlm.Plan plan, int n_frames, int vs_max, "
          "int di_max, int dd_max, bool use_side, bool timed, int lse_cu_share_pct) -> int", decode_frames);
    m.def("decode_batch(__torch__.torch.classes.jlm.Model model, __torch__.torch.classes.jlm.Plan plan, Tensor host_ints, int head_end, "
          "Tensor? blk, int[] blk_src, int[] blk_dst, int[] blk_n, Tensor(a!) h_nodes, Tensor(b!) h_len, Tensor(c!) h_score, Tensor? h_nlive, "
          "int n_frames, int vs_max, int di_max, int dd_max, bool use_side, bool timed, int lse_cu_share_pct) -> int", decode_batch);
    m.def("frame_times(__torch__.torch.classes.jlm.Plan plan) -> Tensor", frame_times);
    m.def("lstm_step(Tensor h_in, Tensor c_in, int ld_state, Tensor(a!) h_out, Tensor(b!) c_out, Tensor? rows, Tensor prev, Tensor word, "
          "Tensor emb, int ld_emb, Tensor wt, Tensor bias, int kpad, int H, int E, int n_rows_max, Tensor? n_dev) -> ()", lstm_step);
    m.def("gemm_nt(Tensor A, int a_off, int lda, Tensor? a_rows, Tensor B, int ldb, Tensor? b_rows, Tensor(a!) C, int c_off, int ldc, "
          "Tensor? c_rows, Tensor? bias, int bias_off, int M, int N, int K, Tensor? m_dev) -> ()", gemm_nt);
    m.def("softmax_rows(Tensor y, Tensor(a!) pred, int ld, int n_rows, int n_cols, bool self_norm) -> ()", softmax_rows);
    m.def("pack_split_f16(Tensor src, int src_off, int rows, int k, int ld, float scale, Tensor(a!) dst, int dst_off, int ld_dst) -> ()",
          pack_split_f16);
    m.def("pack_split_f16_col(Tensor v, int v_off, int rows, float scale, Tensor(a!) dst, int ld_dst, int col) -> ()", pack_split_f16_col);
    m.def("pack_mixed(Tensor src, int src_off, int rows, int k, int ld, Tensor bias, int bias_off, float scale, float bias_scale, float s8, "
          "Tensor(a!) dst, int ld_dst) -> ()", pack_mixed);
    m.def("dequant_u8(Tensor code, int rows, int k, int ld_code, Tensor codebook, Tensor(a!) dst, int ld_dst) -> ()", dequant_u8);
    m.def("lse_probe(__torch__.torch.classes.jlm.Model model, Tensor rowlist, Tensor prev, Tensor word, int steps, int rows, Tensor(a!) h, "
          "Tensor(b!) c, Tensor(c!) T, Tensor? Tm, int ld_tm, int form, Tensor(d!) part, int max_parts) -> int", lse_probe);
    m.def("abi_version() -> int", abi_version);
    m.def("beam_step_max_cands(int beam, int n_frames, int mode) -> int", beam_step_max_cands);
}
