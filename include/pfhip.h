/*
 * pfhip.h — plain-C ABI of the MI355X (gfx950) Paraformer acoustic-model forward path.
 *
 * This is the drop-in boundary beneath the reference's `funasr::Model` plug-in seam
 * (onnxruntime/include/model.h:13-46).  Every entry point names the reference interface it replaces
 * (paths relative to the reference root).  Plain C types only; no exceptions cross this boundary;
 * every function returns a pfhip_status and pfhip_last_error() holds a thread-local message.
 *
 * Threading: like the reference's `Model::Forward`, which all decoder threads call concurrently on
 * one shared handle (websocket/bin/funasr-wss-server.cpp:479-481), every entry point taking a
 * pfhip_model* is re-entrant.  A handle owns ONE weight set per device and one or more execution
 * contexts over it (workspace + streams; pfhip_set_inflight): concurrent pfhip_offline_forward calls
 * run on different contexts — merged into packed launches first when pfhip_set_batching is on — and
 * calls that land on the same context are serialised on its lock.
 */
#ifndef PFHIP_H_
#define PFHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfhip_model pfhip_model;

typedef enum {
  PFHIP_OK = 0,
  PFHIP_ERR_ARG = 1,         /* null / out-of-range argument                                   */
  PFHIP_ERR_HIP = 2,         /* a HIP runtime call failed (message has hipGetErrorString)      */
  PFHIP_ERR_FORMAT = 3,      /* manifest / blob malformed or tensor missing                    */
  PFHIP_ERR_UNSUPPORTED = 4, /* configuration outside what the gfx950 kernels are built for    */
  PFHIP_ERR_CAPACITY = 5     /* caller-provided output buffer too small                        */
} pfhip_status;

/* Thread-local description of the last failure on this thread ("" if none). */
const char* pfhip_last_error(void);

/* ---- model lifetime --------------------------------------------------------------------------
 * Replaces Paraformer::InitAsr(am_model, am_cmvn, am_config, token_file, thread_num)
 * (onnxruntime/src/paraformer.cpp:21-53): loads weights + CMVN + config and builds the fbank tables
 * (knf options of paraformer.cpp:24-31).  Weight container: flat little-endian float32 blob + JSON
 * manifest (see asr-2pass_amd/weights.py).  `device` is the HIP device ordinal (one model replica
 * per GPU; SURVEY.md §8e "replicas only"). */
pfhip_status pfhip_create(const char* weight_blob_path, const char* manifest_json_path, int device,
                          pfhip_model** out);
pfhip_status pfhip_create_from_memory(const void* blob, size_t blob_bytes, const char* manifest_json,
                                      int device, pfhip_model** out);
/* The same with the strings the reference itself passes — its own file contract (onnxruntime/include/com-define.h:52-88):
 *   am_model     <dir>/model.onnx | model_quant.onnx (offline-stream.cpp:74-77, tpass-stream.cpp:62,68) | model.torchscript /
 *                model_blade.torchscript (offline-stream.cpp:79-84: TorchScript archives are not opened, the ONNX file of the same
 *                stem beside it is read) | a container x.pfhip.bin (am_config = its manifest)
 *   second_model NULL, or the online model's decoder.onnx / decoder_quant.onnx (paraformer.cpp:56-121: en_model + de_model)
 *   hw_model     NULL, or model_eb.onnx as handed to InitHwCompiler BEFORE InitAsr (offline-stream.cpp:60-72, paraformer.cpp:243-261)
 *   am_cmvn      am.mvn, read as LoadCmvn does (paraformer.cpp:325-360)
 *   am_config    config.yaml (LoadConfigFromYaml / LoadOnlineConfigFromYaml, paraformer.cpp:178-241)
 * The weights are read by a dependency-free protobuf walk (anonymous transposed MatMul initializers named after their layer,
 * LSTM gate order, dynamic quantisation folded back to float32); nothing in a model file is executed.  The converted container
 * is cached beside the source as <stem>.pfhip.{bin,json} and reused while every source file keeps its size and mtime
 * (PFHIP_MODEL_CACHE=0 switches the cache off; a read-only directory is not an error). */
pfhip_status pfhip_create_from_files(const char* am_model, const char* second_model, const char* hw_model, const char* am_cmvn,
                                     const char* am_config, int device, pfhip_model** out);
/* The file-reading step on its own (no device needed): kind = "asr" | "vad" | "punc".  The blob / manifest pointers stay
 * valid until pfhip_container_free. */
typedef struct pfhip_container pfhip_container;
pfhip_status pfhip_read_model_files(const char* kind, const char* model, const char* second, const char* hotword, const char* cmvn,
                                    const char* config, pfhip_container** out);
const float* pfhip_container_blob(const pfhip_container* c, size_t* bytes);
const char* pfhip_container_manifest(const pfhip_container* c);
int pfhip_container_from_cache(const pfhip_container* c);
void pfhip_container_free(pfhip_container* c);
/* What the wire-format walk sees in ONE .onnx file, as JSON: initializer count and bytes, node count, node inputs nobody
 * produces (0 for a file read correctly), torch-style tensors and the sum of their values.  Inspection only. */
pfhip_status pfhip_onnx_summary(const char* path, char* out, size_t cap);
/* Replaces Paraformer::~Paraformer (paraformer.cpp:267-295). */
void pfhip_destroy(pfhip_model* m);

/* Model facts the host adapter needs (Model::GetAsrSampleRate model.h:40; vocab size = logits width). */
int pfhip_sample_rate(const pfhip_model* m);
int pfhip_vocab_size(const pfhip_model* m);
int pfhip_feat_dim(const pfhip_model* m);   /* lfr_m * n_mels (560) */
int pfhip_d_model(const pfhip_model* m);

/* ---- offline forward --------------------------------------------------------------------------
 * Replaces the body of Paraformer::Forward / ParaformerTorch::Forward between `float** din` and the
 * greedy token ids (onnxruntime/src/paraformer.cpp:463-589; batched contract
 * onnxruntime/src/paraformer-torch.cpp:301-475): FbankKaldi (:309-323) -> LfrCmvn (:421-461) ->
 * `m_session_->Run` (:541) -> GreedySearch/FindMax (:386-395, util.cpp:63-74).
 *
 * Caller-owned output buffers (any optional pointer may be NULL):
 *   token_ids [batch*max_tokens]  argmax ids, row-major per utterance (first-max-wins)
 *   token_num [batch]             floor(sum alphas) — the reference's outputTensor[1] (paraformer.cpp:547)
 *   n_fires   [batch]             rows of log-probs the graph produced for the utterance (CIF fires)
 *   n_frames  [batch]             LFR frames T (paraformer.cpp:483)
 *   logp      [batch*max_tokens*vocab]  log-probs (only when a WFST decoder consumes them,
 *                                 paraformer.cpp:563-579); rows >= n_fires are untouched
 * An utterance shorter than one fbank window yields token_num = n_fires = 0 (paraformer.cpp:477-480).
 */
typedef struct {
  int32_t* token_ids;
  int32_t* token_num;
  int32_t* n_fires;
  int32_t* n_frames;
  float* logp;
  int32_t max_tokens;
  /* Timestamp models only (the reference's 4-output graph, paraformer.cpp:545-562 `outputTensor.size() == 4`): the
   * x3-upsampled alphas and CIF integrate trace that TimestampOnnx consumes (pfhip_timestamp_onnx below), one row of
   * max_us floats per utterance, us_len[b] = 3 * n_frames[b] valid.  Any of the three may be NULL; all NULL = skip. */
  float* us_alphas;
  float* us_peaks;
  int32_t* us_len;
  int32_t max_us;
} pfhip_out;

/* Host-buffer form: pcm[i] points at n_samples[i] floats in [-1,1) exactly as Model::Forward gets
 * them from Audio::FetchDynamic (onnxruntime/src/audio.cpp:1052-1108).  Does H2D, the forward, D2H. */
pfhip_status pfhip_offline_forward(pfhip_model* m, const float* const* pcm, const int* n_samples,
                                   int batch, const float* hw_emb, int n_hotwords, pfhip_out* out);

/* Cross-request batching for the host-buffer form: with wait_us > 0, concurrent pfhip_offline_forward callers (the
 * server's decoder threads, funasr-wss-server.cpp:479-481) are merged into packed forwards of up to max_utterances
 * utterances.  ONE queue per handle feeds every execution slot (contexts, replica devices): the caller at its front claims an
 * idle slot and runs the batch it gathered while the next caller already gathers the next one.  A caller that finds nothing in
 * flight runs at once (a lone caller never waits); while every slot is busy the gathering is free; only with an idle slot AND
 * other batches executing does a leader wait — at most wait_us — for company.  Results are identical to separate calls.
 * 0 switches it off (default).  Contextual models are not merged (hotwords are per connection); calls with at least
 * max_utterances utterances go straight to the least-loaded slot. */
pfhip_status pfhip_set_batching(pfhip_model* m, int wait_us, int max_utterances);

/* Execution contexts: how many offline forwards of this handle may be in flight per device.  The reference shares ONE
 * Ort::Session among all decoder threads (onnxruntime/src/paraformer.cpp:35-41,541; websocket/bin/funasr-wss-server.cpp:479-481);
 * here the threads share one weight set (blob, repacks, LayerNorm-folded copies, front-end tables) and each forward runs on one
 * of n contexts — a workspace, two streams and the state of its last batch, a few KB until its first forward sizes the
 * workspace.  One launch of a large batch fills the chip, but its bandwidth-bound phases (epilogues, FSMN prologue, launch
 * boundaries, the host round trip for the token counts) leave the matrix cores idle; a second and third batch in flight fill
 * those gaps (+8..12 % audio-s/s at 3).  With pfhip_set_batching the merged batches are dealt to idle contexts from ONE queue.
 * n = 1 (default) is one forward at a time.  PFHIP_INFLIGHT=n in the environment applies it to every handle created.
 * Call at initialisation (it may allocate); contexts of a replica group are created on every device.  The device-pointer form
 * (pfhip_offline_enqueue / _fetch), streams, pfhip_profile_* and pfhip_extract_feats use context 0. */
pfhip_status pfhip_set_inflight(pfhip_model* m, int n);
/* Replaces ParaformerTorch::WarmUp, the dummy forward the reference's GPU flavour runs inside InitAsr
 * (onnxruntime/src/paraformer-torch.cpp:59,477-520): one synthetic batch of `batch` utterances x `n_samples` samples through
 * EVERY execution slot (context x device), so that code objects are loaded and each context's workspace is sized before the
 * first request.  Call after pfhip_set_inflight. */
pfhip_status pfhip_warm_up(pfhip_model* m, int batch, int n_samples);
int pfhip_get_inflight(const pfhip_model* m);
/* One entry per execution slot (context x device) of the handle: packed device forwards run there, caller calls and utterances
 * they served.  utterances / forwards > calls / forwards > 1 is cross-request merging at work.  *n_out = slots (also on
 * PFHIP_ERR_CAPACITY). */
typedef struct {
  int32_t device, context;
  int64_t forwards, calls, utterances;
} pfhip_slot_stats;
pfhip_status pfhip_inflight_stats(pfhip_model* m, pfhip_slot_stats* out, int cap, int* n_out);

/* Device-resident form (what bench.py times): d_pcm is ONE device buffer holding the utterances
 * back to back; sample_off/n_samples are host arrays.  All kernels are enqueued on `stream`
 * (a hipStream_t, NULL = the model's own stream); results stay in the model's device workspace until
 * pfhip_offline_fetch.  One host sync happens inside (the CIF token counts size the decoder launch). */
pfhip_status pfhip_offline_enqueue(pfhip_model* m, const float* d_pcm, const int64_t* sample_off,
                                   const int* n_samples, int batch, void* stream);
pfhip_status pfhip_offline_fetch(pfhip_model* m, pfhip_out* out);
/* pfhip_offline_forward for audio that is already in HBM (d_pcm / sample_off / n_samples as pfhip_offline_enqueue; the buffer
 * must be complete and stay untouched until the call returns): the forward and the D2H of the results on the least-loaded
 * execution slot's own stream, synchronous like the host-buffer form.  What bench.py's `value` times from several threads on
 * ONE handle; d_pcm must live on the device of the handle (groups: device 0's slots only would see it — single-device handles). */
pfhip_status pfhip_offline_forward_resident(pfhip_model* m, const float* d_pcm, const int64_t* sample_off, const int* n_samples,
                                            int batch, pfhip_out* out);

/* ---- hotwords (contextual model) ---------------------------------------------------------------
 *   pfhip_hotword_embed <-> the `hw_m_session->Run` on model_eb.onnx + the per-hotword row selection inside
 *                           Paraformer::CompileHotwordEmbedding (paraformer.cpp:656-685): ids i32 [H,10] (0-padded, last row
 *                           [1,0,...] appended by the caller, :648-651) + lengths [H] -> f32 [H, d] = LSTM output at step len-1.
 *                           The string -> id part (:601-647: split, seg_dict, vocabulary) is host text handling above this.
 *   pfhip_set_hotwords  <-> the `hw_emb` argument of Model::Forward kept resident for pfhip_offline_enqueue;
 *                           pfhip_offline_forward takes hw_emb [H, d] per call like the reference (paraformer.cpp:515-531).
 * A contextual model without hotwords is an error ("hw_emb is null", :516-520); a plain model ignores them. */
int pfhip_is_contextual(const pfhip_model* m);
/* 1 when the model carries the CifPredictorV3 upsampling head (the reference's 4-output graph, paraformer.cpp:545). */
int pfhip_has_timestamp_head(const pfhip_model* m);
pfhip_status pfhip_hotword_embed(pfhip_model* m, const int32_t* hotword_matrix, const int32_t* lengths, int n_hotwords,
                                 float* out);
pfhip_status pfhip_set_hotwords(pfhip_model* m, const float* hw_emb, int n_hotwords);

/* ---- front end only ----------------------------------------------------------------------------
 * Replaces Paraformer::FbankKaldi + LfrCmvn (paraformer.cpp:309-323, 421-461) on their own:
 * feats_out gets sum(n_frames)*feat_dim floats, utterances back to back; n_frames_out [batch]. */
pfhip_status pfhip_extract_feats(pfhip_model* m, const float* const* pcm, const int* n_samples,
                                 int batch, float* feats_out, size_t feats_cap_floats,
                                 int32_t* n_frames_out);

/* ---- chunk-streaming forward ---------------------------------------------------------------------
 * One pfhip_stream per connection = one `funasr::ParaformerOnline` (onnxruntime/src/paraformer-online.cpp),
 * created from the ONLINE model's container; its caches (fbank splice cache, [5|10|5] overlap window,
 * CIF carry, 16 decoder FSMN caches) live in HBM.  Calls on streams of one model are serialised.
 *   pfhip_stream_create   <-> ParaformerOnline::ParaformerOnline + InitCache   (:12-62, 347-384)
 *   pfhip_stream_forward  <-> ParaformerOnline::Forward(din, len, input_finished) (:525-601), i.e.
 *                             ExtractFeats/OnlineLfrCmvn (:147-238), x*sqrt(d)+GetPosEmb (:549-555, 240-268),
 *                             AddOverlapChunk (:397-413), ForwardChunk = encoder Run + CifSearch + decoder Run
 *                             + OnlineGreedySearch (:415-523, 270-345; paraformer.cpp:362-371)
 *   pfhip_stream_reset    <-> Reset + ResetCache (:386-395)
 * token_ids receives the ids of the tokens this call emitted (the reference returns their text); at most
 * 32000 samples per call (the 2-pass server sends 9600, websocket-server-2pass.cpp:135-148). */
typedef struct pfhip_stream pfhip_stream;
pfhip_status pfhip_stream_create(pfhip_model* m, const int* chunk_size /* [3] or NULL = {5,10,5} */,
                                 pfhip_stream** out);
void pfhip_stream_destroy(pfhip_stream* s);
pfhip_status pfhip_stream_reset(pfhip_stream* s);
pfhip_status pfhip_stream_forward(pfhip_stream* s, const float* pcm, int n_samples, int input_finished,
                                  int32_t* token_ids, int cap, int* n_tokens);
/* Which branch of ParaformerOnline::Forward the stream's last call took: 0 not a final call (or no feature row), 1 the short
 * final call that flushes the look-back cache (:532-540), 2 a final call whose rows fit one last chunk (:557-559; the only one
 * whose non-empty text the reference ends with a blank, :585-587), 3 first chunk + last chunk (:560-579). */
int pfhip_stream_last_path(const pfhip_stream* s);
/* The same call for n_streams connections of ONE model at once (each stream at most once): every connection runs its own
 * ParaformerOnline::Forward control flow, and the encoder windows that are ready are packed into one forward (a 20-row
 * window costs the full weight stream and ~700 launches whatever its size).  Results are identical to n_streams separate
 * pfhip_stream_forward calls.  token_ids[i] has room for cap[i] ids; n_tokens[i] receives the count.
 * Errors are per connection where they can be: a connection whose cap[i] is too small receives no ids and n_tokens[i] =
 * the count it needed, the call returns PFHIP_ERR_CAPACITY, and every OTHER connection of the batch is served as usual
 * (callers merged by pfhip_set_stream_batching each get their own status).  A failure of the shared forward itself
 * (HIP error, more than 72 CIF fires in one chunk) fails all connections and re-initialises their caches (as
 * pfhip_stream_reset), so no stream is left with a half-advanced chunk. */
pfhip_status pfhip_stream_forward_batch(pfhip_stream* const* streams, int n_streams, const float* const* pcm,
                                        const int* n_samples, const int* input_finished, int32_t* const* token_ids,
                                        const int* cap, int* n_tokens);
/* Cross-connection batching behind the per-connection call: with wait_us > 0, concurrent pfhip_stream_forward callers on
 * different streams of model m (the 2-pass server's one-strand-per-connection threads, websocket-server-2pass.cpp:266-297)
 * are merged into one pfhip_stream_forward_batch of up to max_streams connections; the first caller waits at most wait_us.
 * Results are identical to separate calls.  0 switches it off (default). */
pfhip_status pfhip_set_stream_batching(pfhip_model* m, int wait_us, int max_streams);
/* Inspection of the LAST encoder window of the last call: "chunk" [n,560], "enc" [n,d], "alphas" [n],
 * "emb" [fires,d], "logp" [fires,vocab] (log-probs are only kept after pfhip_stream_set_debug(s,1)).
 * set_debug bit 1 = keep log-probs.  The tensors are read from the model's packed workspace: valid until the next forward. */
pfhip_status pfhip_stream_set_debug(pfhip_stream* s, int on);
pfhip_status pfhip_stream_get_tensor(pfhip_stream* s, const char* name, float* dst, size_t cap_floats,
                                     size_t* n_out);

/* ---- FSMN-VAD forward ---------------------------------------------------------------------------
 *   pfhip_vad_create_from_memory <-> FsmnVad::InitVad (ReadModel + LoadCmvn + InitCache, fsmn-vad.cpp:12-70, 258-263)
 *   pfhip_vad_forward            <-> FsmnVad::FbankKaldi + LfrCmvn + Forward (fsmn-vad.cpp:137-152, 198-238, 72-135):
 *                                    probs [T, n_classes] row-major (class 0 = silence, e2e-vad.h:103); the four
 *                                    [128 x 19] caches stay in HBM and are advanced only when !is_final (:129-134)
 *   pfhip_vad_reset              <-> FsmnVad::InitCache (:258-263) */
typedef struct pfhip_vad pfhip_vad;
pfhip_status pfhip_vad_create_from_memory(const void* blob, size_t blob_bytes, const char* manifest_json, int device,
                                          pfhip_vad** out);
/* FsmnVad::InitVad(vad_model, vad_cmvn, vad_config, thread_num) with the reference's own strings (fsmn-vad.cpp:10-50:
 * <vad-dir>/model.onnx | model_quant.onnx, am.mvn, config.yaml); see pfhip_create_from_files. */
pfhip_status pfhip_vad_create_from_files(const char* vad_model, const char* vad_cmvn, const char* vad_config, int device, pfhip_vad** out);
void pfhip_vad_destroy(pfhip_vad* v);
pfhip_status pfhip_vad_reset(pfhip_vad* v);
int pfhip_vad_num_classes(const pfhip_vad* v);
pfhip_status pfhip_vad_forward(pfhip_vad* v, const float* pcm, int n_samples, int is_final, float* probs,
                               size_t cap_floats, int* n_frames);

/* Same forward, but only the silence posterior (class 0) of each frame comes back — the one column the end-point
 * detector reads (e2e-vad.h:103,607-609): T floats instead of T x 248. */
pfhip_status pfhip_vad_forward_sil(pfhip_vad* v, const float* pcm, int n_samples, int is_final, float* sil_prob,
                                   size_t cap_floats, int* n_frames);

/* ---- online FSMN-VAD: one pfhip_vad_stream per connection = one `funasr::FsmnVadOnline` --------------------------------
 * (onnxruntime/src/fsmn-vad-online.cpp; built on the offline handle like FsmnVadOnline(FsmnVad*), :206-219).
 *   pfhip_vad_stream_infer <-> FsmnVadOnline::Infer up to the scorer (:135-147): ExtractFeats with input_cache_ /
 *                              lfr_splice_cache_ / reserve_waveforms_ (:11-88), OnlineLfrCmvn (:90-133), Forward with this
 *                              connection's caches.  sil_prob receives the silence posterior of the *n_frames rows this call
 *                              produced; waves_out the waveform FsmnVadOnline hands to vad_scorer for the same rows (:148) —
 *                              feed both to pfhip_vadseg_feed(online = 1).  A final call is scored against zeroed caches and
 *                              resets the object, as the reference does (Reset/ResetCache inside ExtractFeats, :84-87).
 *   pfhip_vad_stream_reset <-> Reset + ResetCache (:160-163, fsmn-vad-online.h:59-63) */
typedef struct pfhip_vad_stream pfhip_vad_stream;
pfhip_status pfhip_vad_stream_create(pfhip_vad* v, pfhip_vad_stream** out);
void pfhip_vad_stream_destroy(pfhip_vad_stream* vs);
pfhip_status pfhip_vad_stream_reset(pfhip_vad_stream* vs);
pfhip_status pfhip_vad_stream_infer(pfhip_vad_stream* vs, const float* pcm, int n_samples, int input_finished, float* sil_prob,
                                    size_t cap_floats, int* n_frames, float* waves_out, size_t waves_cap, int* n_waves);

/* The same call for n_streams connections of ONE pfhip_vad at once (arrays of the per-connection arguments above): the 2-pass
 * server's websocket handlers each call FsmnVadOnline::Infer per 600 ms message (websocket-server-2pass.cpp:85-117 ->
 * funasrruntime.cpp:516-532); issued together they share every launch (fbank, OnlineLfrCmvn rows, the 11 GEMMs, the FSMN
 * memory with per-connection caches), so a round of N connections costs about what one call does.  Results are identical to
 * n_streams separate calls.  A stream may appear only once per batch. */
pfhip_status pfhip_vad_stream_infer_batch(pfhip_vad_stream* const* streams, int n_streams, const float* const* pcm,
                                          const int* n_samples, const int* input_finished, float* const* sil_prob,
                                          const size_t* cap_floats, int* n_frames, float* const* waves_out,
                                          const size_t* waves_cap, int* n_waves);

/* Merge concurrent pfhip_vad_stream_infer callers (one thread per connection, as the reference's websocket handlers are) into
 * batched passes: wait_us > 0 and max_streams > 1 make the first caller wait up to wait_us for others.  Default: off. */
pfhip_status pfhip_set_vad_stream_batching(pfhip_vad* v, int wait_us, int max_streams);

/* ---- VAD end-point detector (host logic) -----------------------------------------------------------
 * `funasr::E2EVadModel` (onnxruntime/src/e2e-vad.h:268-783, WindowDetector :181-266, VADXOptions defaults :78-107)
 * restated on the host: sequential threshold / window logic over ~100 frames per second, no device work.
 *   pfhip_vadseg_feed <-> E2EVadModel::operator()(score, waveform, is_final, online, max_end_sil,
 *                         max_single_segment_time, speech_noise_thres, sample_rate) (:303-362); segments are written as
 *                         (start_ms, end_ms) pairs, -1 = "open" in online mode.  *n_segments > cap_pairs -> PFHIP_ERR_CAPACITY. */
typedef struct pfhip_vadseg pfhip_vadseg;
pfhip_status pfhip_vadseg_create(pfhip_vadseg** out);
void pfhip_vadseg_destroy(pfhip_vadseg* s);
pfhip_status pfhip_vadseg_reset(pfhip_vadseg* s);
pfhip_status pfhip_vadseg_feed(pfhip_vadseg* s, const float* sil_prob, int n_frames, const float* waveform,
                               int n_samples, int is_final, int online, int max_end_sil, int max_single_segment_time,
                               float speech_noise_thres, int sample_rate, int32_t* segments, int cap_pairs,
                               int* n_segments);

/* ---- token time stamps (host logic) ------------------------------------------------------------------
 * `funasr::TimestampOnnx` (onnxruntime/src/util.cpp:838-963) restated on the host: us_alphas / us_cif_peak [3T] of the
 * time-stamp model + the number of recognised tokens (without "</s>") -> (begin_s, end_s) spans; spans[3*i+2] != 0 marks
 * an inserted <sil>.  us_alphas is rescaled in place like the reference's by-reference argument. */
pfhip_status pfhip_timestamp_onnx(float* us_alphas, const float* us_cif_peak, int n_frames3, int n_chars, float begin_time_ms,
                                  float total_offset, float* spans, int cap_spans, int* n_spans);
/* `funasr::PostProcess` (onnxruntime/src/util.cpp:720-836), host string handling: n hypothesis tokens (vocabulary strings,
 * UTF-8) + their (begin_s, end_s) stamps -> "<text> | <b0>, <e0>,<b1>, <e1>..." exactly as Paraformer::GreedySearch returns
 * it in time-stamp mode; special tokens dropped, "@@" pieces merged, Latin words spaced.  *n_out = strlen (cap >= +1). */
pfhip_status pfhip_post_process(const char* const* chars, const float* stamps, int n, char* out, int cap, int* n_out);

/* ---- CT-Transformer punctuation forward ------------------------------------------------------------
 *   pfhip_punc_create_from_memory <-> CTTransformer::InitPunc session load (ct-transformer.cpp:14-37)
 *   pfhip_punc_infer              <-> CTTransformer::Infer (ct-transformer.cpp:162-204): ids [n] -> punctuation id
 *                                     per token = first maximum over the first CANDIDATE_NUM-1 classes; logits_out
 *                                     (optional) gets the [n, n_classes] scores.  Tokenisation and the 20-token
 *                                     mini-sentence bookkeeping of AddPunc (:39-155) stay on the host above this. */
typedef struct pfhip_punc pfhip_punc;
pfhip_status pfhip_punc_create_from_memory(const void* blob, size_t blob_bytes, const char* manifest_json, int device,
                                           pfhip_punc** out);
/* CTTransformer::InitPunc(punc_model, punc_config, token_file, thread_num)'s session load with the reference's own strings
 * (ct-transformer.cpp:14-37: <punc-dir>/model.onnx | model_quant.onnx, config.yaml; model_conf.punc_list goes into the manifest's
 * config.punc_list for the host tokenizer); see pfhip_create_from_files. */
pfhip_status pfhip_punc_create_from_files(const char* punc_model, const char* punc_config, int device, pfhip_punc** out);
void pfhip_punc_destroy(pfhip_punc* p);
int pfhip_punc_num_classes(const pfhip_punc* p);
pfhip_status pfhip_punc_infer(pfhip_punc* p, const int32_t* ids, int n, int32_t* punc_out, float* logits_out);
/* Realtime variant <-> CTTransformerOnline::Infer(input_data, nCacheSize) (ct-transformer-online.cpp:154-223): the
 * VadMask of :225-240 (queries before cache_size-1 do not see tokens from cache_size on) is applied in every attention
 * block, because the reference feeds that one mask to both `vad_mask` and `sub_masks` (:182-197).  The model's FSMN
 * look-ahead is the container's `sanm_shift` (0 or 5). */
pfhip_status pfhip_punc_infer_online(pfhip_punc* p, const int32_t* ids, int n, int cache_size, int32_t* punc_out,
                                     float* logits_out);
/* n_seq sequences in one packed device pass (positions, FSMN memory and attention per sequence): results identical to n_seq
 * separate pfhip_punc_infer calls (cache_size == NULL) / pfhip_punc_infer_online calls (cache_size[b] per sequence).
 * pfhip_set_punc_batching(wait_us > 0, max_sequences > 1) merges concurrent single-sequence callers (one AddPunc per handler
 * thread) into such passes; offline and realtime calls are never mixed in one pass.  Default: off. */
pfhip_status pfhip_punc_infer_batch(pfhip_punc* p, const int32_t* const* ids, const int* n, const int* cache_size, int n_seq,
                                    int32_t* const* punc_out);
pfhip_status pfhip_set_punc_batching(pfhip_punc* p, int wait_us, int max_sequences);
/* AddPunc's mini-sentence loop on token ids (ct-transformer.cpp:39-155): 20-token mini-sentences, the tail after the last
 * sentence end carried into the next Infer, forced period at the last comma beyond CACHE_POP_TRIGGER_LIMIT carried tokens,
 * sentence-final fix-up.  punc_out receives one punctuation id per token and, when the text does not end in "。"/"？", one
 * extra PERIOD_INDEX (the reference appends the character): *n_out = n or n + 1 (cap >= n + 1).  The tokeniser and the
 * string assembly stay on the host above this. */
pfhip_status pfhip_punc_add_punc(pfhip_punc* p, const int32_t* ids, int n, int32_t* punc_out, int cap, int* n_out);

/* ---- inspection (parity tests) -----------------------------------------------------------------
 * Copies a named intermediate of the LAST forward to host: "feats" [M,560], "enc" [M,d],
 * "alphas" [M] (without the tail slot), "emb" [sum fires, d], "logp" [sum fires, vocab].
 * Rows are utterance-major in batch order.  *n_out = floats written. */
pfhip_status pfhip_get_tensor(pfhip_model* m, const char* name, float* dst, size_t cap_floats,
                              size_t* n_out);

/* ---- per-kernel timing (bench.py roofline leg) -------------------------------------------------
 * When enabled, each launch of a kernel class is bracketed by hipEvents on the launch stream.
 * `on`: 0 = off, 1 = every class, any other value = bit mask of classes (bit c = class c; bench.py times only
 * the dominant class, 1<<0 | 1<<8 ... see below, so that the event records do not perturb the timed region).
 * Classes: 0 gemm, 1 attention, 2 layernorm, 3 fsmn, 4 fbank, 5 cif, 6 head(log-softmax/argmax), 7 other. */
#define PFHIP_NUM_KCLASS 8
typedef struct {
  double ms[PFHIP_NUM_KCLASS];       /* summed device time                                         */
  int64_t launches[PFHIP_NUM_KCLASS];
  double flops[PFHIP_NUM_KCLASS];    /* algorithmic flops issued (pad rows/cols excluded)           */
  double bytes[PFHIP_NUM_KCLASS];    /* algorithmic bytes (compulsory reads + writes)               */
} pfhip_profile;
/* ---- in-process multi-GPU: one handle, one replica per device (replicas only, no collective) -----------------
 * The reference server holds ONE model handle that all `decoder-thread-num` threads call into (funasr-wss-server.cpp:479-481,
 * websocket-server.cpp:387-403).  A group keeps that shape on a multi-GPU node: the handle returned is replica 0, further
 * replicas (full copies of the weights) live on the other devices; pfhip_offline_forward routes each call (or each merged
 * batch, pfhip_set_batching) to the replica with the fewest calls in flight, pfhip_stream_create pins a new connection to the
 * replica with the fewest open streams for its lifetime (device-resident caches), pfhip_set_hotwords / pfhip_set_batching /
 * pfhip_set_stream_batching apply to every replica.  Results do not depend on which replica served a call.  No data moves
 * between devices.  For several forwards in flight on ONE device use pfhip_set_inflight (contexts share the weights); listing a
 * device twice here still works but builds a second full copy of the weights there.
 * PFHIP_DEVICES="0,1,..." in the environment makes pfhip_create / pfhip_create_from_memory build such a group (their `device`
 * argument is then ignored), so the stock server needs no code change to use every GPU of the node.
 * pfhip_offline_enqueue / pfhip_offline_fetch (device pointers) always use replica 0.  pfhip_group_stats counts per replica
 * (its contexts included); pfhip_inflight_stats per execution slot. */
pfhip_status pfhip_create_group(const void* blob, size_t blob_bytes, const char* manifest_json, const int* devices, int n_devices,
                                pfhip_model** out);
int pfhip_group_size(const pfhip_model* m);
/* Per replica (arrays of at least pfhip_group_size entries, any may be NULL): device ordinal, offline calls and utterances
 * served so far, streams open now. */
pfhip_status pfhip_group_stats(pfhip_model* m, int* devices, int64_t* calls, int64_t* utterances, int* open_streams, int cap);

/* Test hook, not part of the serving path: "blstm_flag" (value != 0) raises the timestamp head's error word for the next
 * timestamp request only — as a step-barrier time-out of the persistent BLSTM kernel would — which then has to be served by the
 * per-step recurrence; "blstm_fallbacks" returns (as the status value) how many requests were served that way; "plane_forwards"
 * how many forwards of this context ran the encoder on plane-image operands (gemm_p3.hip: batches of PFHIP_PLANES_MIN_ROWS+ rows). */
pfhip_status pfhip_debug_poke(pfhip_model* m, const char* what, int value);
pfhip_status pfhip_profile_enable(pfhip_model* m, int on);
pfhip_status pfhip_profile_read(pfhip_model* m, pfhip_profile* out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PFHIP_H_ */
