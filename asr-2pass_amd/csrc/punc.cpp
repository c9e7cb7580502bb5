// CT-Transformer punctuation forward on MI355X — the C ABI's pfhip_punc_* family.
// Replaces the `m_session->Run` + Argmax loop of CTTransformer::Infer (onnxruntime/src/ct-transformer.cpp:162-204):
// token ids [N] -> logits [N, 6] -> punctuation id = first maximum over the first CANDIDATE_NUM-1 classes.
// Same kernels as the ASR encoder (fp32 MFMA GEMMs, FSMN memory block, fused attention instantiated for
// d_k = 32) plus an embedding gather fused with the x*sqrt(d) + sinusoidal PE step.  The 272727 x 256
// embedding table (279 MB) stays in HBM; a call touches N rows of it.
#include <memory>

#include "internal.h"
#include "json_min.h"

using namespace pfhip_detail;

namespace pfhip {
void launch_embed_gather(const int32_t* ids, const float* table, int vocab, int D, float* out, int ldo, int N,
                         const float* inv_ts, float scale, const int* pos_of_row, hipStream_t s);   // punc.hip
void launch_argmax_first(const float* logits, int ldl, int N, int ncls, int32_t* out, hipStream_t s);
}

struct PuncReq : pfhip_detail::MergeReqBase {
  const int32_t* ids; int n; int vad_pos; int32_t* out;
  pfhip_status st = PFHIP_OK; std::string err;
};

struct pfhip_punc {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  int vocab = 0, d = 256, n_head = 8, ffn = 1024, layers = 4, n_punc = 6, sanm_shift = 0;
  float* d_table = nullptr;
  float* d_inv_ts = nullptr;
  struct Layer { float *n1g, *n1b, *n2g, *n2b, *fsmn; Lin qkv, out, ffn1, ffn2; };
  std::vector<Layer> L;
  float *an_g = nullptr, *an_b = nullptr;
  Lin head;
  std::vector<float*> owned;
  Buf ids, x, y, qkv, mem, ctx, h, logits, punc, meta, lim;
  int* h_pin = nullptr;
  // pinned staging for one (batched) call: [ids | pos | lim | off | len] in, punctuation ids out
  int* h_stage = nullptr; size_t h_stage_cap = 0;
  int* stage(size_t n_ints) {
    if (n_ints > h_stage_cap) {
      if (h_stage) (void)hipHostFree(h_stage);
      h_stage = nullptr; h_stage_cap = 0;
      const size_t cap = n_ints + n_ints / 2 + 1024;
      if (hipHostMalloc((void**)&h_stage, cap * 4, hipHostMallocDefault) == hipSuccess) h_stage_cap = cap;
    }
    return h_stage;
  }
  // merging of concurrent callers (pfhip_set_punc_batching)
  pfhip_detail::MergeQueue<PuncReq> mq;
  int q_wait_us = 0, q_max = 1;
};

extern "C" {

pfhip_status pfhip_punc_create_from_memory(const void* blob, size_t blob_bytes, const char* manifest_json, int device,
                                           pfhip_punc** out) {
  last_error().clear();
  if (!blob || !manifest_json || !out) return fail(PFHIP_ERR_ARG, "null argument");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(PFHIP_ERR_ARG, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  pfhip::JValue man;
  try { man = pfhip::JParser(manifest_json).parse(); }
  catch (const std::exception& e) { return fail(PFHIP_ERR_FORMAT, e.what()); }
  const pfhip::JValue* jc = man.get("config");
  const pfhip::JValue* jt = man.get("tensors");
  if (!jc || !jt || jt->kind != pfhip::JValue::OBJ) return fail(PFHIP_ERR_FORMAT, "manifest needs config and tensors");
  std::unique_ptr<pfhip_punc> p(new pfhip_punc);
  p->device = device;
  p->vocab = (int)jc->number("vocab", 0);
  p->d = (int)jc->number("d_model", 256);
  p->n_head = (int)jc->number("n_head", 8);
  p->ffn = (int)jc->number("ffn", 1024);
  p->layers = (int)jc->number("layers", 4);
  p->n_punc = (int)jc->number("n_punc", 6);
  const int kernel = (int)jc->number("kernel", 11);
  p->sanm_shift = (int)jc->number("sanm_shift", 0);
  if (p->sanm_shift != 0 && p->sanm_shift != 5) return fail(PFHIP_ERR_UNSUPPORTED, "sanm_shift must be 0 or 5");
  const int d = p->d;
  if (p->vocab <= 0 || d % 128 || d > 512 || d / p->n_head != 32 || kernel != 11 || p->ffn % 128 || p->ffn > 2048 ||
      p->n_punc < 2 || p->n_punc > 128)
    return fail(PFHIP_ERR_UNSUPPORTED, "CT-Transformer kernels need d_model/n_head == 32, FSMN kernel 11");
  const float* hb = static_cast<const float*>(blob);
  auto get = [&](const std::string& name, std::vector<int> shape, const float** ptr) -> bool {
    const pfhip::JValue* t = jt->get(name);
    if (!t) { last_error() = "missing tensor " + name; return false; }
    const pfhip::JValue* sh = t->get("shape");
    const pfhip::JValue* of = t->get("offset");
    if (!sh || !of || sh->arr.size() != shape.size()) { last_error() = "tensor " + name + " malformed"; return false; }
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); ++i) {
      if ((int)sh->arr[i].num != shape[i]) { last_error() = "tensor " + name + " has unexpected shape"; return false; }
      n *= (size_t)shape[i];
    }
    const size_t off = (size_t)of->num;
    if (off % 4 || off + n * 4 > blob_bytes) { last_error() = "tensor " + name + " out of blob"; return false; }
    *ptr = hb + off / 4;
    return true;
  };
  auto dev_copy = [&](const float* src, size_t n, float** dst) -> pfhip_status {
    HIP_TRY(hipMalloc((void**)dst, n * 4));
    HIP_TRY(hipMemcpy(*dst, src, n * 4, hipMemcpyHostToDevice));
    p->owned.push_back(*dst);
    return PFHIP_OK;
  };
  auto vec = [&](const std::string& name, int n, float** dst) -> pfhip_status {
    const float* src = nullptr;
    if (!get(name, {n}, &src)) return PFHIP_ERR_FORMAT;
    return dev_copy(src, n, dst);
  };
  auto lin = [&](const std::string& name, int N, int K, Lin* l) -> pfhip_status {
    const float *w = nullptr, *b = nullptr;
    if (!get(name + ".w", {N, K}, &w) || !get(name + ".b", {N}, &b)) return PFHIP_ERR_FORMAT;
    return pack_linear(w, b, N, K, l);
  };
  pfhip_status st;
  {
    const float* tab = nullptr;
    if (!get("embed.w", {p->vocab, d}, &tab)) return PFHIP_ERR_FORMAT;
    if ((st = dev_copy(tab, (size_t)p->vocab * d, &p->d_table))) return st;
  }
  p->L.resize(p->layers);
  for (int i = 0; i < p->layers; ++i) {
    const std::string q = "enc." + std::to_string(i) + ".";
    pfhip_punc::Layer& l = p->L[i];
    const float* fw = nullptr;
    if ((st = vec(q + "norm1.g", d, &l.n1g)) || (st = vec(q + "norm1.b", d, &l.n1b)) || (st = vec(q + "norm2.g", d, &l.n2g)) ||
        (st = vec(q + "norm2.b", d, &l.n2b)) || (st = lin(q + "qkv", 3 * d, d, &l.qkv)) || (st = lin(q + "out", d, d, &l.out)) ||
        (st = lin(q + "ffn1", p->ffn, d, &l.ffn1)) || (st = lin(q + "ffn2", d, p->ffn, &l.ffn2)))
      return st;
    if (!get(q + "fsmn.w", {d, kernel}, &fw)) return PFHIP_ERR_FORMAT;
    if ((st = dev_copy(fw, (size_t)d * kernel, &l.fsmn))) return st;
  }
  if ((st = vec("enc.after_norm.g", d, &p->an_g)) || (st = vec("enc.after_norm.b", d, &p->an_b)) ||
      (st = lin("out", p->n_punc, d, &p->head)))
    return st;
  {
    const int half = d / 2;
    std::vector<float> inv(half);
    const float scale = (float)(-std::log(10000.0) / (half - 1));
    for (int i = 0; i < half; ++i) inv[i] = (float)exp((double)(i * scale));
    if ((st = dev_copy(inv.data(), half, &p->d_inv_ts))) return st;
  }
  HIP_TRY(p->meta.ensure(256));
  HIP_TRY(hipHostMalloc((void**)&p->h_pin, 256, hipHostMallocDefault));
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  *out = p.release();
  return PFHIP_OK;
}

void pfhip_punc_destroy(pfhip_punc* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  for (Buf* b : {&p->ids, &p->x, &p->y, &p->qkv, &p->mem, &p->ctx, &p->h, &p->logits, &p->punc, &p->meta, &p->lim}) b->release();
  for (auto& l : p->L) { free_lin(l.qkv); free_lin(l.out); free_lin(l.ffn1); free_lin(l.ffn2); }
  free_lin(p->head);
  for (float* q : p->owned) (void)hipFree(q);
  if (p->h_pin) (void)hipHostFree(p->h_pin);
  if (p->h_stage) (void)hipHostFree(p->h_stage);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

int pfhip_punc_num_classes(const pfhip_punc* p) { return p ? p->n_punc : 0; }

// B sequences packed row-wise through one pass: per-sequence positions for the embedding, (offset, length) pairs for the FSMN
// memory and the attention, per-row key limits for the realtime mask.  vad_pos == nullptr: the offline model.
static pfhip_status punc_infer_packed(pfhip_punc* p, const int32_t* const* ids, const int* n, const int* vad_pos, int B,
                                      int32_t* const* punc_out, float* logits_out);

static pfhip_status punc_infer_queued(pfhip_punc* p, const int32_t* ids, int n, int vad_pos, int32_t* punc_out);

pfhip_status pfhip_punc_infer(pfhip_punc* p, const int32_t* ids, int n, int32_t* punc_out, float* logits_out) {
  last_error().clear();
  if (!p || !ids || n <= 0 || !punc_out) return fail(PFHIP_ERR_ARG, "bad argument");
  if (p->q_wait_us > 0 && p->q_max > 1 && !logits_out) return punc_infer_queued(p, ids, n, -1, punc_out);
  return punc_infer_packed(p, &ids, &n, nullptr, 1, &punc_out, logits_out);
}

pfhip_status pfhip_punc_infer_online(pfhip_punc* p, const int32_t* ids, int n, int cache_size, int32_t* punc_out,
                                     float* logits_out) {
  last_error().clear();
  if (!p || !ids || n <= 0 || !punc_out) return fail(PFHIP_ERR_ARG, "bad argument");
  const int vp = cache_size < 0 ? 0 : cache_size;
  if (p->q_wait_us > 0 && p->q_max > 1 && !logits_out) return punc_infer_queued(p, ids, n, vp, punc_out);
  return punc_infer_packed(p, &ids, &n, &vp, 1, &punc_out, logits_out);
}

pfhip_status pfhip_set_punc_batching(pfhip_punc* p, int wait_us, int max_sequences) {
  last_error().clear();
  if (!p || wait_us < 0 || max_sequences < 1) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> l(p->mq.mu);
  p->q_wait_us = wait_us;
  p->q_max = max_sequences;
  return PFHIP_OK;
}

pfhip_status pfhip_punc_infer_batch(pfhip_punc* p, const int32_t* const* ids, const int* n, const int* cache_size, int n_seq,
                                    int32_t* const* punc_out) {
  last_error().clear();
  if (!p || !ids || !n || n_seq <= 0 || !punc_out) return fail(PFHIP_ERR_ARG, "bad argument");
  for (int b = 0; b < n_seq; ++b)
    if (!ids[b] || n[b] <= 0 || !punc_out[b]) return fail(PFHIP_ERR_ARG, "bad sequence");
  std::vector<int> vp;
  if (cache_size) { vp.assign(cache_size, cache_size + n_seq); for (int& v : vp) v = v < 0 ? 0 : v; }
  return punc_infer_packed(p, ids, n, cache_size ? vp.data() : nullptr, n_seq, punc_out, nullptr);
}

// One Infer per handler thread (every connection's AddPunc makes a few): merged into packed passes like the ASR calls.
static pfhip_status punc_infer_queued(pfhip_punc* p, const int32_t* ids, int n, int vad_pos, int32_t* punc_out) {
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= p->vocab) return fail(PFHIP_ERR_ARG, "token id outside the embedding table");
  PuncReq me;
  me.ids = ids; me.n = n; me.vad_pos = vad_pos; me.out = punc_out;
  int wait_us, cap;
  { std::lock_guard<std::mutex> l(p->mq.mu); wait_us = p->q_wait_us; cap = p->q_max; }
  p->mq.submit(
      me, wait_us, [&](const std::deque<PuncReq*>& q) { return (int)q.size() >= cap; },
      [&](std::deque<PuncReq*>& q, std::vector<PuncReq*>& take) {
        // one pass serves one kind of mask: offline (vad_pos < 0) and realtime calls are not mixed
        const bool online = q.front()->vad_pos >= 0;
        std::deque<PuncReq*> later;
        while (!q.empty() && (int)take.size() < cap) {
          PuncReq* r = q.front();
          q.pop_front();
          if ((r->vad_pos >= 0) == online) take.push_back(r); else later.push_back(r);
        }
        for (auto it = later.rbegin(); it != later.rend(); ++it) q.push_front(*it);
      },
      [&](std::vector<PuncReq*>& take) {
        const int B = (int)take.size();
        std::vector<const int32_t*> ids_v(B);
        std::vector<int> n_v(B), vp(B);
        std::vector<int32_t*> out_v(B);
        for (int b = 0; b < B; ++b) { ids_v[b] = take[b]->ids; n_v[b] = take[b]->n; vp[b] = take[b]->vad_pos; out_v[b] = take[b]->out; }
        const pfhip_status st = punc_infer_packed(p, ids_v.data(), n_v.data(), take[0]->vad_pos >= 0 ? vp.data() : nullptr, B,
                                                  out_v.data(), nullptr);
        const std::string err = last_error();
        for (PuncReq* r : take) { r->st = st; r->err = err; }
      },
      /*fresh_no_wait=*/true);          // a lone caller runs at once; company gathers behind the pass that is executing
  if (me.st != PFHIP_OK) last_error() = me.err;
  return me.st;
}

static pfhip_status punc_infer_packed(pfhip_punc* p, const int32_t* const* ids, const int* n, const int* vad_pos, int B,
                                      int32_t* const* punc_out, float* logits_out) {
  int total = 0, max_n = 0;
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < n[b]; ++i)
      if (ids[b][i] < 0 || ids[b][i] >= p->vocab) return fail(PFHIP_ERR_ARG, "token id outside the embedding table");
    total += n[b];
    max_n = std::max(max_n, n[b]);
  }
  std::lock_guard<std::mutex> lk(p->mu);
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int d = p->d, Np = round_up(total, 128);
  HIP_TRY(p->x.ensure((size_t)Np * d * 4));
  HIP_TRY(p->y.ensure((size_t)Np * d * 4));
  HIP_TRY(p->qkv.ensure((size_t)Np * 3 * d * 4));
  HIP_TRY(p->mem.ensure((size_t)Np * d * 4));
  HIP_TRY(p->ctx.ensure((size_t)Np * d * 4));
  HIP_TRY(p->h.ensure((size_t)Np * p->ffn * 4));
  HIP_TRY(p->logits.ensure((size_t)Np * 128 * 4));
  HIP_TRY(p->punc.ensure((size_t)total * 4));
  // one upload: [ids | pos | lim | off | len]
  const size_t n_in = (size_t)3 * total + (size_t)2 * B;
  int* hs = p->stage(n_in + (size_t)total);
  if (!hs) return fail(PFHIP_ERR_HIP, "pinned staging allocation failed");
  int *h_ids = hs, *h_pos = hs + total, *h_lim = hs + 2 * total, *h_off = hs + 3 * total, *h_len = h_off + B;
  int* h_out = hs + n_in;
  for (int b = 0, r = 0; b < B; ++b) {
    h_off[b] = r; h_len[b] = n[b];
    std::memcpy(h_ids + r, ids[b], (size_t)n[b] * 4);
    // CTTransformerOnline::VadMask (ct-transformer-online.cpp:225-240) as a per-query key limit: rows i < vad_pos-1 see
    // keys [0, vad_pos), all other rows see everything; the reference feeds this ONE mask to both mask inputs (:182-197)
    const int vp = vad_pos ? vad_pos[b] : 0;
    for (int i = 0; i < n[b]; ++i) {
      h_pos[r + i] = i;
      h_lim[r + i] = (vp > 0 && vp < n[b] && i < vp - 1) ? vp : n[b];
    }
    r += n[b];
  }
  HIP_TRY(p->meta.ensure(n_in * 4));
  HIP_TRY(hipMemcpyAsync(p->meta.p, hs, n_in * 4, hipMemcpyHostToDevice, s));
  const int* d_ids = p->meta.i();
  const int* d_pos = d_ids + total;
  const int* d_lim = vad_pos ? d_ids + 2 * total : nullptr;
  const int* d_off = d_ids + 3 * total;
  const int* d_len = d_off + B;
  float* x = p->x.f();
  pfhip::launch_embed_gather(d_ids, p->d_table, p->vocab, d, x, d, total, p->d_inv_ts, sqrtf((float)d), B > 1 ? d_pos : nullptr, s);
  const float att_scale = 1.0f / sqrtf(32.f);
  for (int i = 0; i < p->layers; ++i) {
    const pfhip_punc::Layer& l = p->L[i];
    pfhip::launch_layernorm(x, d, p->y.f(), d, l.n1g, l.n1b, total, d, d, 1e-12f, s);
    lin_gemm(s, l.qkv, p->y.f(), d, p->qkv.f(), 3 * d, nullptr, 0, nullptr, 0, total, false);
    pfhip::launch_fsmn_shift(p->qkv.f() + 2 * d, 3 * d, l.fsmn, nullptr, 0, p->mem.f(), d, d_off, d_len, B, max_n, d,
                             p->sanm_shift, s);
    pfhip::launch_attention_masked(p->qkv.f(), 3 * d, p->qkv.f() + d, 3 * d, p->qkv.f() + 2 * d, 3 * d, p->ctx.f(), d, d_off,
                                   d_len, d_off, d_len, d_lim, B, p->n_head, max_n, att_scale, 32, s);
    lin_gemm(s, l.out, p->ctx.f(), d, x, d, p->mem.f(), d, x, d, total, false);          // in_size == size: residual
    pfhip::launch_layernorm(x, d, p->y.f(), d, l.n2g, l.n2b, total, d, d, 1e-12f, s);
    lin_gemm(s, l.ffn1, p->y.f(), d, p->h.f(), p->ffn, nullptr, 0, nullptr, 0, total, true);
    lin_gemm(s, l.ffn2, p->h.f(), p->ffn, x, d, x, d, nullptr, 0, total, false);
  }
  pfhip::launch_layernorm(x, d, p->y.f(), d, p->an_g, p->an_b, total, d, d, 1e-12f, s);
  lin_gemm(s, p->head, p->y.f(), d, p->logits.f(), 128, nullptr, 0, nullptr, 0, total, false);
  // Argmax(p, p + CANDIDATE_NUM - 1): first maximum over the first n_punc-1 classes (ct-transformer.cpp:193-196)
  pfhip::launch_argmax_first(p->logits.f(), 128, total, p->n_punc - 1, static_cast<int32_t*>(p->punc.p), s);
  HIP_TRY(hipMemcpyAsync(h_out, p->punc.p, (size_t)total * 4, hipMemcpyDeviceToHost, s));
  if (logits_out)
    HIP_TRY(hipMemcpy2DAsync(logits_out, (size_t)p->n_punc * 4, p->logits.p, 128 * 4, (size_t)p->n_punc * 4, total,
                             hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  for (int b = 0; b < B; ++b) std::memcpy(punc_out[b], h_out + h_off[b], (size_t)n[b] * 4);
  return PFHIP_OK;
}

}  // extern "C"

// ---- AddPunc's mini-sentence bookkeeping on token ids (ct-transformer.cpp:39-155) --------------------
// The text side of AddPunc (tokeniser, joining words and punctuation strings, en-bpe symbol mapping) stays above this call;
// what is restated here decides WHICH ids each Infer sees and which punctuation every token finally gets: 20-token
// mini-sentences, the un-punctuated tail of a mini-sentence carried into the next one (RemainIDs), a forced period at the
// last comma once the carried text exceeds CACHE_POP_TRIGGER_LIMIT, and the sentence-final fix-up of the last token.
pfhip_status pfhip_punc_add_punc(pfhip_punc* p, const int32_t* ids, int n, int32_t* punc_out, int cap, int* n_out) {
  last_error().clear();
  if (!p || !ids || n < 0 || !punc_out || !n_out) return fail(PFHIP_ERR_ARG, "bad argument");
  constexpr int kTokenLen = 20, kCachePopTriggerLimit = 200;           // com-define.h:126,136
  constexpr int kNotPunc = 1, kComma = 2, kPeriod = 3, kQuestion = 4, kDun = 5;   // com-define.h:131-135
  *n_out = 0;
  std::vector<int32_t> remain, out;
  const int total_batch = (n + kTokenLen - 1) / kTokenLen;
  for (int i = 0; i < n; i += kTokenLen) {
    const int take = std::min(kTokenLen, n - i);
    std::vector<int32_t> input(remain);
    input.insert(input.end(), ids + i, ids + i + take);
    std::vector<int32_t> punc(input.size());
    pfhip_status st = pfhip_punc_infer(p, input.data(), (int)input.size(), punc.data(), nullptr);
    if (st) return st;
    const int cur_batch = i / kTokenLen;
    if (cur_batch < total_batch - 1) {                                // not the last mini-sentence (:66-90)
      int sent_end = -1, last_comma = -1;
      for (int k = (int)punc.size() - 2; k > 0; --k) {
        if (punc[k] == kPeriod || punc[k] == kQuestion) { sent_end = k; break; }
        if (last_comma < 0 && punc[k] == kComma) last_comma = k;
      }
      if (sent_end < 0 && (int)input.size() > kCachePopTriggerLimit && last_comma > 0) {
        sent_end = last_comma;
        punc[sent_end] = kPeriod;
      }
      remain.assign(input.begin() + (sent_end + 1), input.end());
      punc.resize((size_t)(sent_end + 1));
    }
    out.insert(out.end(), punc.begin(), punc.end());
  }
  if (!out.empty()) {                                                 // last mini-sentence (:112-127)
    const int last = out.back();
    if (last == kComma || last == kDun) out.back() = kPeriod;
    else if (last != kPeriod && last != kQuestion) out.push_back(kPeriod);   // a period is APPENDED after the last token
  }
  (void)kNotPunc;
  if ((int)out.size() > cap) return fail(PFHIP_ERR_CAPACITY, "punc_out too small (needs up to n + 1)");
  std::copy(out.begin(), out.end(), punc_out);
  *n_out = (int)out.size();
  return PFHIP_OK;
}
