/* pk2hip.h -- C ABI of libpk2hip.so, the MI355X (gfx950) implementation of the
 * PyKaldi2 sequence-training hot path.
 *
 * The reference (jzlianglu/pykaldi2) has no FFI of its own: its hot path is
 * Python calling torch.nn / PyKaldi / Horovod.  Each entry point below replaces
 * one of those third-party calls; the reference call site is cited.  The
 * boundary is plain C: raw device pointers, sizes, a hipStream_t passed as
 * void*.  No torch types.  All device memory (inputs, outputs, workspaces) is
 * allocated and owned by the caller; the library never allocates device memory
 * on the hot path (graph handles own their static arrays, created once).
 *
 * Every function returns 0 on success or a negative pk2_status; the message of
 * the last failure on the calling thread is at pk2_last_error().  Kernels are
 * enqueued on `stream`; nothing synchronises unless stated.
 */
#ifndef PK2HIP_H_
#define PK2HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  PK2_OK = 0,
  PK2_ERR_INVALID = -1,   /* bad argument */
  PK2_ERR_HIP = -2,       /* HIP runtime error (see pk2_last_error) */
  PK2_ERR_LIMIT = -3,     /* size exceeds a documented kernel limit */
  PK2_ERR_IO = -4,        /* file could not be read / bad format */
  PK2_ERR_NUMERIC = -5    /* a device computation failed (e.g. the decoder lost every token) */
} pk2_status;

const char* pk2_last_error(void);
int pk2_version(void);               /* ABI version, currently 1 */

/* ------------------------------------------------------------------ *
 * Denominator graph.  Replaces kaldi.chain.DenominatorGraph(den_fst, P)
 * (reference bin/train_chain.py:167,202).  Host arrays describe den.fst's
 * arcs: prob = exp(-weight), pdf = ilabel-1.  The library computes
 * initial_probs (100-iteration rule), builds its three arc orderings
 * (by destination / by source / by pdf) and uploads them once.
 * ------------------------------------------------------------------ */
typedef struct pk2_den_graph pk2_den_graph;

int pk2_den_graph_create(int32_t num_states, int32_t num_pdfs, int64_t num_arcs,
                         const int32_t* arc_src, const int32_t* arc_dst,
                         const int32_t* arc_pdf, const float* arc_prob,
                         int32_t start_state, pk2_den_graph** out);
/* Reads an OpenFst binary StdVectorFst (den.fst) -- StdVectorFst.read at
 * reference bin/train_chain.py:167. */
int pk2_den_graph_from_openfst(const char* path, int32_t num_pdfs, pk2_den_graph** out);
int pk2_den_graph_destroy(pk2_den_graph* g);
int pk2_den_graph_info(const pk2_den_graph* g, int32_t* num_states, int32_t* num_pdfs,
                       int64_t* num_arcs);
/* Copies initial_probs (num_states floats) to a host buffer. */
int pk2_den_graph_initial_probs(const pk2_den_graph* g, float* host_out);

/* Which kernels a denominator call on num_seqs sequences takes on the current device: 0 = per-arc-pdf kernels,
 * 1 = state-x kernels, one launch per frame, 2 = state-x, persistent recursion kernel (one launch per call).  Reporting
 * only; the choice is the library's (reference: kaldi.chain.DenominatorComputation behind ops/ops.py:265). */
int32_t pk2_den_graph_path(const pk2_den_graph* g, int32_t num_seqs);
/* The form of the persistent recursion kernel behind path 2: 1 = every arc in registers and the whole state vector in LDS
 * (graphs up to ~1.05 M arc slots / ~36 k states), 2 = chunked LDS table whose copy overlaps the arcs, two resident passes
 * and streamed overflow (any graph with < 65536 states); 0 = none (paths 0 / 1).  PK2_DEN_PERSIST = 0 | 1 | 2 forces one. */
int32_t pk2_den_graph_persist_form(const pk2_den_graph* g, int32_t num_seqs);

/* ------------------------------------------------------------------ *
 * LF-MMI objective and derivative for a minibatch of N sequences.
 * Replaces kaldi.chain.compute_chain_objf_and_deriv (reference ops/ops.py:265)
 * plus the wrapper's own grad += xent_regularize * grad_xent (ops/ops.py:267),
 * without the host round trips of ops/ops.py:255,261,269-271.
 *
 * logits: device f32; sequence n, frame t, pdf p at
 *         logits[n*seq_stride + t*frame_stride + p]   (frame_stride >= P).
 * lengths: host int32[N], frames_per_sequence of each supervision.
 * grad:   device f32, same addressing as logits (grad_seq_stride/frame_stride);
 *         receives d objf / d logits for t < lengths[n] and 0 for the padding
 *         frames lengths[n] <= t < max_frames.
 * Numerator supervisions (Kaldi chain::Supervision FSTs: acyclic, one frame
 * per arc, label = pdf) are concatenated over the batch, on the device:
 *   num_arc_src/dst/pdf int32[total_arcs], num_arc_weight f32 (-log prob),
 *   arcs of one sequence sorted by the frame of their source state;
 *   num_frame_off int32[sum(lengths)+N]: for sequence n (base = sum_{m<n}
 *   (lengths[m]+1)) entry base+t is the first arc (global index) leaving frame
 *   t, entry base+lengths[n] is one past the last;
 *   num_state_off int32[N+1] gives each sequence's state-id base (arc src/dst
 *   are local to the sequence); state 0 is initial;
 *   num_final_state int32[total_finals], num_final_weight f32,
 *   num_final_off int32[N+1].
 * out (device f32[3*N]): objf[n] = weight*(log p_num - log p_den),
 *   num_logprob[n], den_logprob[n].  Kaldi's guard is mirrored: if a value is
 *   not finite or the alpha-beta check fails, objf[n] = -10*weight*lengths[n]
 *   and that sequence's gradient is zero.
 * workspace: device, at least pk2_chain_workspace_bytes() bytes.
 * ------------------------------------------------------------------ */
typedef struct {
  const int32_t* arc_src;
  const int32_t* arc_dst;
  const int32_t* arc_pdf;
  const float* arc_weight;
  const int32_t* frame_off;
  const int32_t* state_off;     /* host pointer, int32[N+1] */
  const int32_t* final_state;
  const float* final_weight;
  const int32_t* final_off;     /* host pointer, int32[N+1] */
  int64_t total_arcs;
  int64_t max_seq_arcs;         /* arcs of the largest single sequence (0 = unknown: total_arcs is the bound).  Decides
                                 * whether a sequence's frame table and arcs fit the LDS of its workgroup (the staged
                                 * forward-backward, which also rides inside the persistent denominator launch). */
} pk2_num_batch;

size_t pk2_chain_workspace_bytes(const pk2_den_graph* g, int32_t num_seqs, int32_t max_frames,
                                 int64_t num_total_states);

int pk2_chain_objf_and_deriv(const pk2_den_graph* g, const float* logits, int64_t seq_stride,
                             int64_t frame_stride, const int32_t* lengths, int32_t num_seqs,
                             const pk2_num_batch* num, float leaky_hmm_coefficient,
                             float xent_regularize, float l2_regularize, float weight,
                             float* grad, int64_t grad_seq_stride, int64_t grad_frame_stride,
                             float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The same computation in the convention of the reference's autograd operator (ops/ops.py:243-280: forward returns the
 * objective, backward returns MINUS its derivative whatever grad_out is): the gradient written is
 * grad_scale * d objf / d logits (grad_scale = -1 for the operator) and *objf_sum (device f32, optional) receives
 * sum_n objf[n] -- so that the operator needs no elementwise or reduction kernel of its own around the call. */
int pk2_chain_objf_and_deriv_op(const pk2_den_graph* g, const float* logits, int64_t seq_stride,
                                int64_t frame_stride, const int32_t* lengths, int32_t num_seqs,
                                const pk2_num_batch* num, float leaky_hmm_coefficient,
                                float xent_regularize, float l2_regularize, float weight,
                                float* grad, int64_t grad_seq_stride, int64_t grad_frame_stride,
                                float* out, void* workspace, size_t workspace_bytes, float grad_scale,
                                float* objf_sum, void* stream);

/* Denominator only (log p_den and its occupancies); used by tests/profiling.
 * den_logprob: device f32[N]; gamma: device, logits addressing, receives
 * d log p_den / d logits (posteriors, rows sum to 1). */
int pk2_chain_den_fwd_bwd(const pk2_den_graph* g, const float* logits, int64_t seq_stride,
                          int64_t frame_stride, const int32_t* lengths, int32_t num_seqs,
                          float leaky_hmm_coefficient, float* den_logprob, float* gamma,
                          int64_t gamma_seq_stride, int64_t gamma_frame_stride, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Test hook: the objective / guard rule of pk2_chain_objf_and_deriv (Kaldi ComputeChainObjfAndDeriv: objf finite and
 * |alpha-beta product - 1| <= 2.0, else objf = -10 * weight * frames and a zero derivative) applied to given device arrays
 * of N per-sequence quantities.  out: device f32[3N] = {objf, num_lp, den_lp}; flags: device i32[N]. */
int pk2_chain_debug_flags(const float* num_lp, const float* den_lp, const float* alpha_beta_product,
                          const int32_t* lengths, int32_t N, float weight, float* out, int32_t* flags, void* stream);

/* ------------------------------------------------------------------ *
 * Chain supervision of one utterance from its alignment (host-side integer work, no device memory).
 * Replaces, for bin/train_chain.py:262-272 of the reference,
 *   aligner.to_phone_alignment(trans_ids)                       (Kaldi SplitToPhones),
 *   kaldi.chain.alignment_to_proto_supervision(opts, phones, durations),
 *   kaldi.chain.proto_supervision_to_supervision(tree, trans_model, proto, convert_to_pdfs=True).
 * Kaldi composes the phone sequence with the context and H transducers, adds self-loops (reordered) and
 * composes with a time-enforcing FST; the frame-synchronous acceptor those operations define is built
 * here directly: state = (frames consumed, phone instance, HMM state of the last frame), see
 * csrc/chain_sup.hip.  Arc weights are 0 (Kaldi builds H with transition_scale = 0 and adds the self-loops
 * with self_loop_scale = 0).
 * ------------------------------------------------------------------ */
/* Splits a transition-id alignment into phones.  Tables are indexed by transition-id (1-based, entry 0
 * unused): tid_tstate = transition-state, tid_phone, tid_flags bit 0 = self-loop, bit 1 = transition to the
 * final HMM state.  phones / durations: capacity T.  Returns PK2_ERR_INVALID for a transition-id out of
 * range; *ok = 0 when the alignment is not a complete sequence of phones (Kaldi's SplitToPhones returns
 * false: the pieces are still produced). */
int pk2_split_to_phones(const int32_t* tid_tstate, const int32_t* tid_phone, const uint8_t* tid_flags,
                        int32_t num_tids, const int32_t* alignment, int32_t T, int32_t* phones,
                        int32_t* durations, int32_t* num_phones, int32_t* ok);

/* HMM topologies + context-dependency tree of the chain model (0.trans_mdl and tree of the reference's
 * -chain_dir).
 * Topology: phone2entry[max_phone+1] (-1 = phone without topology); entry e owns the HMM states
 *   entry_state_off[e] .. entry_state_off[e+1]; per state forward / self-loop pdf-class (-1 = non-emitting
 *   final state) and its transitions trans_dst[state_trans_off[s] .. state_trans_off[s+1]) given as HMM
 *   state indices inside the entry.
 * Tuples (optional, num_tuples = 0 skips the check): {phone, hmm_state, forward_pdf, self_loop_pdf} rows of
 *   the transition model; a (phone, state, pdfs) combination the tree produces that is not among them is an
 *   error, as in TransitionModel::TupleToTransitionState.
 * Tree: context width N, central position P, and the EventMap flattened to nodes
 *   kind 0 (constant): a = pdf-id;
 *   kind 1 (table):    key, children = pool[a .. a+b) (node index, -1 = null);
 *   kind 2 (split):    key, yes-set = pool[a .. a+b) sorted ascending, yes child = pool[a+b], no child = pool[a+b+1];
 *   keys: -1 = pdf-class, 0..N-1 = phone at that window position.  Node 0 is the root. */
typedef struct pk2_sup_model pk2_sup_model;
pk2_sup_model* pk2_sup_model_create(int32_t max_phone, const int32_t* phone2entry, int32_t num_entries,
                                    const int32_t* entry_state_off, const int32_t* state_fwd_class,
                                    const int32_t* state_loop_class, const int32_t* state_trans_off,
                                    const int32_t* trans_dst, int32_t num_tuples, const int32_t* tuples,
                                    int32_t context_width, int32_t central_position, int32_t num_nodes,
                                    const int32_t* node_kind, const int32_t* node_key, const int32_t* node_a,
                                    const int32_t* node_b, int32_t pool_size, const int32_t* pool);
void pk2_sup_model_destroy(pk2_sup_model* m);
/* ContextDependency::Compute: window[N] phones (0 = beyond the utterance) -> pdf-id; PK2_ERR_INVALID when the
 * tree has no answer. */
int pk2_sup_model_pdf(const pk2_sup_model* m, const int32_t* window, int32_t pdf_class, int32_t* pdf);

/* AlignmentToProtoSupervision + ProtoSupervisionToSupervision (convert_to_pdfs = true).
 * frames = ceil(sum(durations) / subsampling); phone i spanning frames [b, e) of the alignment may be emitted
 * on subsampled frames ceil(max(0, b - left_tolerance) / f) .. ceil(min(T, e + right_tolerance) / f) - 1;
 * the per-frame test is on the phone identity (TimeEnforcerFst).  Returns null (pk2_last_error) when no
 * path satisfies the constraints or a phone has no topology / tree answer / tuple. */
typedef struct pk2_supervision pk2_supervision;
pk2_supervision* pk2_supervision_create(const pk2_sup_model* m, const int32_t* phones,
                                        const int32_t* durations, int32_t num_phones, int32_t subsampling,
                                        int32_t left_tolerance, int32_t right_tolerance);
void pk2_supervision_destroy(pk2_supervision* s);
void pk2_supervision_sizes(const pk2_supervision* s, int32_t* frames, int32_t* num_states, int32_t* num_arcs,
                           int32_t* num_final, int32_t* num_allowed);
/* Arrays in the layout of pk2_num_batch for one sequence: arcs sorted by the frame of their source state,
 * frame_off[frames+1], states numbered in time order (state 0 initial), state_time[num_states].
 * allowed_off[frames+1] / allowed_phones: the ProtoSupervision's sorted allowed-phone sets.  Null pointers
 * are skipped. */
int pk2_supervision_copy(const pk2_supervision* s, int32_t* arc_src, int32_t* arc_dst, int32_t* arc_pdf,
                         float* arc_weight, int32_t* frame_off, int32_t* state_time, int32_t* final_state,
                         float* final_weight, int32_t* allowed_off, int32_t* allowed_phones);

/* ------------------------------------------------------------------ *
 * Reverberation + additive noise of one utterance (single channel), device arrays throughout.
 * Replaces the numpy arithmetic of the reference's dynamic data simulation: Distorter.apply_rir /
 * Distorter.add_noise (reference simulation/_distorter.py:86-154) as _Simulator.simulate uses them for one
 * speech source (simulation/simulation.py:55-178, called from data/sr_dataset.py:321-345).  The random draws
 * (SNR, noise position) are made by the host and passed in.
 * ------------------------------------------------------------------ */
/* out[i] = (rir * wav)[delay - 1 + i], i in [0, n): apply_rir with sync = True (:147-148); delay = argmax(rir)
 * (delay = 0 keeps the direct path at sample 0).  out must not alias wav. */
int pk2_sim_apply_rir(const float* wav, int64_t n, const float* rir, int32_t k, int32_t delay, float* out,
                      void* stream);
/* stats (device f64[2], zeroed by the caller): stats[0] += sum x^2, stats[1] = max(stats[1], max |x|). */
int pk2_sim_power(const float* x, int64_t n, double* stats, void* stream);
/* mixed[i] += scale * noise_placed[i], scale = sqrt(Px / Pn * 10^(-snr_db / 10)) with Px = sig_stats[0] / n and
 * Pn = noise_stats[0] / m (_comp_noise_scale_given_snr :28-32); 'sample_noise' placement (:36-58): a noise not
 * longer than the signal sits at [start, start + m), a longer one is cropped from `start`. */
int pk2_sim_add_noise(float* mixed, int64_t n, const float* noise, int64_t m, int64_t start, float snr_db,
                      const double* sig_stats, const double* noise_stats, void* stream);
/* x *= 0.5 / stats[1]  (gain normalisation, simulation/simulation.py:170-172). */
int pk2_sim_gain_norm(float* x, int64_t n, const double* stats, void* stream);

/* ------------------------------------------------------------------ *
 * 80-dim log-mel filterbank + CMN + frame subsampling.
 * Replaces DataGeneratorTrain._logfbank_extractor (reference
 * data/sr_dataset.py:279-296 over simulation/freq_analysis.py:113-150),
 * preprocess.cmn(axis=0) (reader/preprocess.py:34-41, called
 * data/sr_dataset.py:365-366) and the roll+unfold subsampling of
 * bin/train_chain.py:251-255.
 * ------------------------------------------------------------------ */
/* mel: host f32[80*257], the rows of data/mel80_window.txt (un-normalised);
 * the column normalisation of sr_dataset.py:283-286 is applied inside. */
typedef struct pk2_fbank pk2_fbank;
int pk2_fbank_create(const float* mel_80x257, pk2_fbank** out);
int pk2_fbank_destroy(pk2_fbank* fb);
/* wav: device f32, utterance n at wav[wav_off[n] .. wav_off[n+1]) (HOST int64
 * offsets).  feats: device f32 [sum_n frames[n]][80] packed in utterance order;
 * frames[n] = pk2_fbank_num_frames(samples).  feat_row_off: DEVICE int64
 * [num_utts+1] prefix sums of frames[] (row of each utterance's first frame).
 * If apply_cmn, the per-utterance mean over time is subtracted. */
int32_t pk2_fbank_num_frames(int64_t num_samples);
int pk2_fbank_compute(const pk2_fbank* fb, const float* wav, const int64_t* wav_off,
                      int32_t num_utts, float* feats, const int64_t* feat_row_off,
                      int32_t apply_cmn, void* stream);
/* Zero-pad to max_t, roll by `shift` (<=0, wraps like torch.roll) and keep
 * every `subsample`-th frame: x(n, j)[:] = feats_n[(j*subsample - shift) mod
 * max_t] (zero where that index >= frames[n]).  x is [num_utts][out_t][80], or
 * [out_t][num_utts][80] when time_major != 0 (the layout the LSTM kernels use). */
int pk2_pad_roll_subsample(const float* feats, const int64_t* feat_row_off, int32_t num_utts,
                           int32_t max_t, int32_t shift, int32_t subsample, float* x,
                           int32_t out_t, int32_t time_major, void* stream);

/* y = (x - mean) / std per feature dimension: GlobalMeanVarianceNormalization.apply_on_ndarray
 * (reference reader/preprocess.py:211-229, applied at data/sr_dataset.py:184-185).  x, y: device f32
 * [rows][dim]; mean, std: device f32 [dim]. */
int pk2_mvn_apply(const float* x, const float* mean, const float* stdv, int64_t rows, int32_t dim,
                  float* y, void* stream);

/* ------------------------------------------------------------------ *
 * Fused log-softmax + NLL + gradient.  Replaces nn.CrossEntropyLoss
 * (ignore_index=-100; reduction mean: reference bin/train_ce.py:134,189;
 * reduction sum: bin/train_se.py:214,235).
 * loss_sum, count: device f32[1] / int32[1] accumulators (zeroed inside);
 * grad receives d(sum loss)/d logits * grad_scale (rows with ignore_index get
 * zeros); the caller divides by count for the mean reduction.
 * ------------------------------------------------------------------ */
int pk2_softmax_ce_fwd_bwd(const float* logits, int64_t row_stride, const int64_t* targets,
                           int64_t ignore_index, int64_t rows, int32_t num_classes,
                           float* loss_sum, int32_t* count, float* grad, int64_t grad_row_stride,
                           float* logprob_out /* optional [rows][P] or NULL */, void* stream);
/* nn.CrossEntropyLoss(reduction='mean') in two launches with NO scaling pass over [rows, classes]: a one-workgroup launch
 * counts the valid targets into *count first, then the kernel above writes grad = d(mean loss)/d logits directly
 * (divided by max(1, count)); loss_sum stays the SUM of the row losses (reference bin/train_ce.py:134,189). */
int pk2_softmax_ce_fwd_bwd_mean(const float* logits, int64_t row_stride, const int64_t* targets,
                                int64_t ignore_index, int64_t rows, int32_t num_classes,
                                float* loss_sum, int32_t* count, float* grad, int64_t grad_row_stride, void* stream);
/* data[i] *= (*num_dev) / (*den_dev) IN PLACE (den_dev may be NULL = 1); when the factor is exactly 1 nothing is read or
 * written beyond the two scalars: the incoming gradient of loss.backward() (1.0) costs no pass over the gradient. */
int pk2_scale_inplace_ratio(float* data, int64_t n, const float* num_dev, const float* den_dev, void* stream);
/* grad *= (*scale_num) / max(1, *count_den) on the device (mean reduction). */
int pk2_scale_by_count(float* data, int64_t n, float numerator, const int32_t* count_den,
                       void* stream);
/* out[i] = in[i] * (*mul_dev) / max(1, *count_den_dev) in one pass (either pointer may be NULL = factor 1; out may equal in):
 * the gradient of nn.CrossEntropyLoss(reduction='mean') times the loss's incoming gradient, both factors on the device
 * (reference bin/train_ce.py:189-195: loss = criterion(...); loss.backward()). */
int pk2_scale_by_scalars(const float* in, float* out, int64_t n, const float* mul_dev, const int32_t* count_den_dev,
                         void* stream);

/* ------------------------------------------------------------------ *
 * f32 MFMA GEMM: C[M,N] (+)= alpha * op(A) * op(B) (+ bias[N]).
 * Replaces the cuBLAS calls under nn.Linear / nn.LSTM (reference
 * models/lstm.py:45-59).  Row-major; lda/ldb/ldc are row strides.
 * transa: A is stored [K,M];  transb: B is stored [N,K] (torch Linear weight).
 * ------------------------------------------------------------------ */
int pk2_gemm_f32(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                 int64_t ldc, const float* bias, void* stream);
/* The product with the elementwise pass behind it in its epilogue: C = act(alpha op(A) op(B) + beta C + bias).
 * act 0: none (= pk2_gemm_f32); 1: ReLU (reference models/transformer.py:60, the encoder layer's
 * linear1 -> F.relu); 2: an element is kept where gate[m*ldg + n] > 0 and 0 elsewhere -- the ReLU's backward mask
 * read from the forward activation (what autograd's threshold_backward does behind the dX product).  Never cut
 * into atomically added K slices. */
int pk2_gemm_f32_act(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, float alpha,
                     const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                     int64_t ldc, const float* bias, int32_t act, const float* gate, int64_t ldg, void* stream);
/* Several products into one set of accumulators, one launch, fixed summation order:
 *   C = act(alpha * sum_{j < nseg} op(A + j*segA) * op(B + j*segB) + beta*C + bias),   K = depth of ONE segment.
 * TransformerAM's Conv1d(kernel 3, padding 1) over time (reference models/transformer.py:88-92: conv1d -> F.relu)
 * is nseg = 3 over row-shifted views of a zero-padded [B + T*B + B, C] activation buffer (segA = +-B*C floats,
 * segB = C*C: the taps' weight slices); segA / segB may be negative; transa = 0 only; act / gate as in
 * pk2_gemm_f32_act. */
int pk2_gemm_f32_seg(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, int32_t nseg, float alpha,
                     const float* A, int64_t lda, int64_t segA, const float* B, int64_t ldb, int64_t segB, float beta,
                     float* C, int64_t ldc, const float* bias, int32_t act, const float* gate, int64_t ldg,
                     void* stream);
/* Batched form: matrix triple z = i0*n1 + i1 (i0 < n0, i1 < n1) lives at offsets i0*stride?0 + i1*stride?1
 * (in floats) from A / B / C; no bias.  Used for the per-(utterance, head) attention products of
 * TransformerAM (reference models/transformer.py:60, nn.MultiheadAttention). */
int pk2_gemm_f32_batched(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, float alpha,
                         const float* A, int64_t lda, int64_t strideA0, int64_t strideA1, const float* B,
                         int64_t ldb, int64_t strideB0, int64_t strideB1, float beta, float* C, int64_t ldc,
                         int64_t strideC0, int64_t strideC1, int32_t n0, int32_t n1, void* stream);
/* Arithmetic of pk2_gemm_f32 / pk2_gemm_f32_batched (process-wide; default from PK2_GEMM_ARITH = bf16x3 | f32):
 *   0 = "f32":    v_mfma_f32_32x32x2_f32, bit-for-bit a k-ordered f32 fmaf chain (157 TFLOP/s peak);
 *   1 = "bf16x3": every f32 operand element is split exactly into three bf16 numbers and six of the nine part products
 *                 run on v_mfma_f32_32x32x16_bf16 with f32 accumulation (csrc/gemm_bf16x3.h): error against float64 no
 *                 larger than the f32 product's (tests/test_gpu_frontend_nn.py::test_gemm_f32_matches_float64 runs both at
 *                 one bound), 2.67 x the matrix-core rate.  Same inputs, same outputs, same memory traffic. */
int pk2_gemm_set_arith(int32_t arith);
int pk2_gemm_get_arith(void);
/* A Linear layer's weight AND bias gradient from one launch: C[M,N] = alpha * A^T B + beta * C with A stored [K,M] (= dY
 * [frames, out_features]), B [K,N] (= the layer's input), and colsum[m] += sum_k A[k][m] (the bias gradient, ACCUMULATED:
 * the buffer holds the gradient so far).  Replaces pk2_gemm_f32(1, 0, ...) followed by pk2_colsum_f32(A, ..., beta = 1). */
int pk2_gemm_f32_tn_colsum(int32_t M, int32_t N, int32_t K, float alpha, const float* A, int64_t lda, const float* B,
                           int64_t ldb, float beta, float* C, int64_t ldc, float* colsum, void* stream);
/* out[n] (+)= sum_m A[m][n]  (bias gradients). */
int pk2_colsum_f32(const float* A, int64_t lda, int32_t M, int32_t N, float beta, float* out,
                   void* stream);

/* ------------------------------------------------------------------ *
 * One (bi)directional LSTM layer, forward and backward through time.
 * Replaces the cuDNN RNN under nn.LSTM (reference models/lstm.py:49-58): gate
 * order i,f,g,o; h0 = c0 = 0; runs over all T frames of every sequence
 * (padding included, as the reference does).
 *
 * All activations are TIME-MAJOR (row = t*B + b) so that "previous step" is a
 * constant row offset and the W_hh gradient is one GEMM over shifted slices.
 * gx:  device f32 [T][B][D*4H], input projections x W_ih^T + b_ih of
 *      direction d at column offset d*4H (computed with pk2_gemm_f32).
 * whh: device f32 [D][4H][H] (weight_hh_l{k}, weight_hh_l{k}_reverse).
 * bhh: device f32 [D][4H] (bias_hh_l{k}[_reverse]) added inside, or NULL.
 * y:   device f32 [T][B][D*H] layer output (h_t; direction d at column d*H).
 * gates: device f32 [D][T][B][4H] post-activation i,f,g,o (saved for backward).
 * cells: device f32 [D][T][B][H] c_t (saved for backward).
 * H in {64,128,256,512,1024}.
 * ------------------------------------------------------------------ */
/* workspace: device f32, pk2_lstm_fwd_workspace_floats(B,H,D) floats (0 for small batches: may be NULL).
 * For B >= 32 the recurrence is tiled as a GEMM with W_hh rows regrouped per hidden unit in it. */
size_t pk2_lstm_fwd_workspace_floats(int32_t B, int32_t H, int32_t num_dirs);
int pk2_lstm_layer_fwd(const float* gx, const float* whh, const float* bhh, int32_t B, int32_t T,
                       int32_t H, int32_t num_dirs, float* y, float* gates, float* cells,
                       void* workspace, void* stream);
/* dy: device f32 [T][B][D*H] gradient wrt y.  dgx: device f32 [T][B][D*4H]
 * receives the gradient wrt the pre-activations (= gradient wrt gx).
 * scratch: device f32, at least pk2_lstm_bwd_scratch_floats(B,H,D). */
/* Batches of <= 4 rows at H = 512 run the whole recurrence of a layer in ONE persistent launch (csrc/lstm_persist.hip;
 * PK2_LSTM_PERSIST=0 keeps the launch-per-step kernels).  A poll of that kernel that times out leaves NaNs in the
 * outputs and raises this flag (the call synchronises the device). */
int pk2_lstm_persist_status(uint32_t* abort_flag);
size_t pk2_lstm_bwd_scratch_floats(int32_t B, int32_t H, int32_t num_dirs);
int pk2_lstm_layer_bwd(const float* dy, const float* whh, const float* gates, const float* cells,
                       int32_t B, int32_t T, int32_t H, int32_t num_dirs, float* dgx,
                       float* scratch, void* stream);
/* The same with the bias gradients where the kernel can produce them on the way: dbias_ih / dbias_hh (device f32
 * [D][4H], may be NULL) are ACCUMULATED into (+=; the caller zeroes them) with the sum over frames and sequences of dgx --
 * b_ih and b_hh of torch's LSTM receive the same gradient.  *bias_done = 1 when they were filled, 0 when the path taken
 * cannot (the caller then calls pk2_colsum_f32 on dgx). */
int pk2_lstm_layer_bwd_bias(const float* dy, const float* whh, const float* gates, const float* cells,
                            int32_t B, int32_t T, int32_t H, int32_t num_dirs, float* dgx, float* scratch,
                            float* dbias_ih, float* dbias_hh, int32_t* bias_done, void* stream);
/* The W_hh gradient dwhh[d] = sum_t dgates[d][t]^T h[d][t-1] (t+1 for the reverse direction)
 * is one pk2_gemm_f32 by the caller over row-shifted slices of dgx and y. */

/* ------------------------------------------------------------------ *
 * Row-wise pieces of TransformerAM (reference models/transformer.py:52-94: nn.TransformerEncoderLayer
 * post-norm + ReLU FFN, Conv1d(k=3)+ReLU per layer, final LayerNorm).
 * ------------------------------------------------------------------ */
/* s = x + res (res may be NULL); y = LayerNorm(s) * gamma + beta.  sum_out (may be NULL) receives s; mean,
 * rstd [rows] are saved for the backward pass. */
int pk2_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int64_t rows,
                      int32_t C, float eps, float* sum_out, float* y, float* mean, float* rstd, void* stream);
/* ds = gradient wrt s; dgamma / dbeta are ACCUMULATED (+=). */
int pk2_layernorm_bwd(const float* dy, const float* s, const float* mean, const float* rstd,
                      const float* gamma, int64_t rows, int32_t C, float* ds, float* dgamma, float* dbeta,
                      void* stream);
/* scores [B*H][T][T] <- softmax over the last axis of (scores + src_mask[T][T]) with columns j where
 * key_padding[b][j] != 0 set to -inf, in place (src_mask / key_padding may be NULL). */
int pk2_softmax_mask_fwd(float* scores, const float* src_mask, const uint8_t* key_padding, int32_t B, int32_t H,
                         int32_t T, void* stream);
/* dP <- P * (dP - rowsum(dP * P)), in place. */
int pk2_softmax_bwd(const float* P, float* dP, int32_t BH, int32_t T, void* stream);

/* Fused multi-head self-attention (head size 64): ctx = softmax(Q K^T * scale + src_mask, padded keys -> -inf) V, read
 * from / written to the packed time-major projections qkv[T*B][3*H*64] (row = t*B + b; Q | K | V) and ctx[T*B][H*64];
 * lse[B*H][T] = log-sum-exp of every score row, kept for the backward pass; dropout on the probabilities with the
 * counter-based mask of pk2_dropout_f32 indexed as a [B*H][T][T] matrix.  The scores never reach HBM.  Replaces the
 * scaled-dot-product core of nn.MultiheadAttention under nn.TransformerEncoderLayer (reference
 * models/transformer.py:52-67); PK2_ERR_INVALID for other head sizes (callers keep the batched-GEMM form for those). */
int pk2_attention_fwd(const float* qkv, int32_t T, int32_t B, int32_t H, int32_t head_dim, float scale,
                      const float* src_mask, const uint8_t* key_padding, float dropout_p, uint64_t seed, float* ctx,
                      float* lse, void* stream);
/* Gradient of the above by recomputation: dqkv[T*B][3*H*64] (every element written) from qkv, ctx, d ctx and lse;
 * dsum[B*H][T] is scratch (rowsum(dctx * ctx)). */
int pk2_attention_bwd(const float* qkv, const float* ctx, const float* dctx, const float* lse, int32_t T, int32_t B,
                      int32_t H, int32_t head_dim, float scale, const float* src_mask, const uint8_t* key_padding,
                      float dropout_p, uint64_t seed, float* dqkv, float* dsum, void* stream);
int pk2_relu_fwd(float* x, int64_t n, void* stream);                 /* in place */
int pk2_relu_bwd(const float* y, float* dy, int64_t n, void* stream); /* dy <- dy * (y > 0) */
int pk2_add_inplace(float* a, const float* b, int64_t n, void* stream);

/* Inverted dropout y = x * m/(1-p), m ~ Bernoulli(1-p), with a counter-based mask that is a pure
 * function of (seed, element index): calling it again on the gradient with the same seed applies the
 * same mask (no mask storage).  Replaces nn.LSTM's inter-layer dropout (reference models/lstm.py:49-54,
 * configs/ce.yaml:19).  x == y allowed. */
int pk2_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------ *
 * Optimiser: global-norm clip + Adam(amsgrad) / SGD(momentum) over one flat
 * f32 parameter buffer.  Replaces clip_grad_norm_ + torch.optim.Adam / SGD
 * (reference bin/train_ce.py:123,195; bin/train_chain.py:138,287-288;
 * bin/train_se.py:127).
 * ------------------------------------------------------------------ */
/* norm_out: device f32[1] receives ||grad||_2. */
int pk2_grad_norm(const float* grad, int64_t n, float* norm_out, void* workspace,
                  size_t workspace_bytes, void* stream);
size_t pk2_grad_norm_workspace_bytes(int64_t n);
/* clip coefficient = min(1, max_norm / (*norm + 1e-6)) read on the device;
 * max_norm <= 0 disables clipping.  step is 1-based.  grad_scale multiplies the
 * gradient first (1/world_size for the averaged all-reduce). */
int pk2_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  float* max_exp_avg_sq /* NULL = no amsgrad */, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float max_norm,
                  const float* norm, float grad_scale, void* stream);
int pk2_sgd_step(float* param, const float* grad, float* momentum_buf /* NULL = none */,
                 int64_t n, float lr, float momentum, float weight_decay, int32_t first_step,
                 float max_norm, const float* norm, float grad_scale, void* stream);

/* ------------------------------------------------------------------ *
 * Guard of the persistent kernels (no reference counterpart: cuDNN / Kaldi kernels cannot time out).
 * The one-launch recurrences, the persistent denominator and the persistent lattice decoder poll each other's results
 * with a 1 s time-out; a launch that gave up has its output poisoned with NaN by a check kernel, which also raises this
 * per-device guard.  While it is raised pk2_adam_step / pk2_sgd_step leave parameters and moments untouched (so that a
 * poisoned gradient never reaches the weights, e.g. in bin/train_ce.py, which has no Kaldi-style NaN guard), and
 * pk2_persist_guard_status reports it WITHOUT synchronising (host-mapped word): pykaldi2_amd.optim raises at the next
 * step().  pk2_persist_guard_clear lowers it (synchronises); pk2_persist_guard_raise raises it from the device (tests).
 * ------------------------------------------------------------------ */
int pk2_persist_guard_status(uint32_t* raised);
int pk2_persist_guard_clear(void);
int pk2_persist_guard_raise(void* stream);
/* Several ranks (hvd.DistributedOptimizer, reference bin/train_chain.py:141-145): the guard is COLLECTIVE.  Before the
 * gradients are exchanged every rank exports its guard word into a one-float device slot, the slots are combined over the
 * ranks (pk2_allreduce_guarded below, or any max / sum all-reduce), and pk2_persist_guard_import raises the local guard
 * when any rank's was raised -- so no rank's optimiser kernel applies a gradient that holds a failed rank's NaN -- and
 * publishes the verdict of step `stamp` (1, 2, ...) in host-mapped memory.  pk2_persist_guard_verdict reads it without
 * synchronising; every rank reads the SAME verdict for the same stamp, so all ranks stop at the same step. */
int pk2_persist_guard_export(float* slot, void* stream);
int pk2_persist_guard_import(const float* slot, uint32_t stamp, void* stream);
int pk2_persist_guard_verdict(uint32_t stamp, uint32_t* ready, uint32_t* raised);

/* ------------------------------------------------------------------ *
 * Lattice path: on-the-fly lattice generation + lattice forward-backward for the MMI / sMBR / MPFE
 * criteria.  Replaces, for a whole minibatch and on the device, what MMIFunction / sMBRFunction do per
 * utterance on the CPU through PyKaldi (reference ops/ops.py:55-66, 133-146):
 *   asr_decoder.decode(loglikes)  [Kaldi LatticeFasterDecoder over HCLG, determinize_lattice = False,
 *                                  reference bin/train_se.py:172-183]
 *   scale_lattice + lattice_forward_backward_mmi(trans_model, lat, trans_ids, True, False, True)
 *   lattice_forward_backward_mpe_variants(trans_model, silence_phones, lat, trans_ids, criterion, True)
 *   Posterior.to_pdf_matrix(trans_model)
 * One workgroup per utterance; the lattices stay in the caller's workspace (device memory).
 * ------------------------------------------------------------------ */
typedef struct pk2_decode_graph pk2_decode_graph;

/* HCLG as arc arrays: ilabel = transition-id (0 = epsilon), weight = tropical cost, final_cost[s] = +inf for
 * non-final states.  Output labels (words) are not needed by any criterion and are not stored. */
int pk2_decode_graph_create(int32_t num_states, int32_t start_state, int64_t num_arcs,
                            const int32_t* arc_src, const int32_t* arc_dst, const int32_t* arc_ilabel,
                            const float* arc_weight, const float* final_cost, pk2_decode_graph** out);
/* HCLG.fst in OpenFst binary form ("vector" or "const" container, "standard" arcs;
 * reference bin/train_se.py:145,180). */
int pk2_decode_graph_from_openfst(const char* path, pk2_decode_graph** out);
/* The same with output labels (word ids; pk2_decode_graph_from_openfst keeps the olabels of the file), and the word of
 * the HCLG arc behind each link of an exported lattice (host arrays; -1 = no such arc): what the lattice-dumping command
 * line needs for `decoder_out["text"]` and the word labels of the compact lattice (reference bin/latgen.py:170-181). */
int pk2_decode_graph_create_words(int32_t num_states, int32_t start_state, int64_t num_arcs,
                                  const int32_t* arc_src, const int32_t* arc_dst, const int32_t* arc_ilabel,
                                  const int32_t* arc_olabel, const float* arc_weight, const float* final_cost,
                                  pk2_decode_graph** out);
int pk2_decode_graph_link_words(const pk2_decode_graph* g, int64_t num_links, const int32_t* src_state,
                                const int32_t* dst_state, const int32_t* tid, const float* graph_cost,
                                int32_t* word_out);
int pk2_decode_graph_destroy(pk2_decode_graph* g);
int pk2_decode_graph_info(const pk2_decode_graph* g, int32_t* num_states, int64_t* num_arcs,
                          int32_t* max_ilabel);

/* LatticeFasterDecoderOptions (reference bin/train_se.py:173-178; YAML decoder_config) + pool sizing. */
typedef struct {
  float beam;             /* 13 */
  float lattice_beam;     /* 7 */
  float beam_delta;       /* 0.5 */
  float acoustic_scale;   /* 0.1 */
  int32_t max_active;     /* 7000 */
  int32_t min_active;     /* 200 */
  int32_t tokens_per_frame;  /* pool size per utterance = (T+1) * this; 0 = max_active */
  int32_t links_per_frame;   /* pool size per utterance = (T+1) * this; 0 = 3 * max_active */
  /* (the Python wrapper passes 2 * max_active / 6 * max_active and quadruples them after an overflow report) */
} pk2_decoder_opts;

/* Layout of one minibatch of lattices (host object; lengths = frames per utterance). */
typedef struct pk2_lattice_batch pk2_lattice_batch;
int pk2_lattice_batch_create(const pk2_decode_graph* g, const int32_t* lengths_host, int32_t num_seq,
                             const pk2_decoder_opts* opts, pk2_lattice_batch** out);
size_t pk2_lattice_batch_bytes(const pk2_lattice_batch* b);
int pk2_lattice_batch_destroy(pk2_lattice_batch* b);

/* Decodes all utterances: loglikes[n][t][pdf] at n*seq_stride + t*frame_stride + pdf (device f32, the
 * acoustic model output minus the log prior, reference bin/train_se.py:241); tid2pdf: device i32
 * [num_tids + 1] (index 0 unused).  Asynchronous on `stream`; the lattices are left in `workspace`
 * (pk2_lattice_batch_bytes bytes, device). */
int pk2_lattice_decode(pk2_lattice_batch* b, const float* loglikes, int64_t seq_stride,
                       int64_t frame_stride, int32_t num_pdfs, const int32_t* tid2pdf, int32_t num_tids,
                       void* workspace, void* stream);
/* By default all frames of an utterance are decoded inside ONE persistent launch by a team of PK2_LAT_TEAM (8 / 16 / 32)
 * workgroups of one XCD (csrc/lattice_decode_frames.hip; PK2_LAT_DECODER=frames: a few launches per frame, =wg: one
 * workgroup per utterance).  *state: 1 = in use on this device, 0 = disabled or failed its first (host-verified) launch
 * -- the launch-per-frame decoder has taken over --, -1 = not tried yet; *abort_flag: some later launch timed out (its
 * utterances were reported with status 5, "not decoded").  Synchronises the device. */
int pk2_lattice_persist_status(int32_t* state, uint32_t* abort_flag);
/* Synchronises `stream` and reports per utterance: status (0 ok, 1 token pool overflow, 2 link pool
 * overflow, 3 no surviving token, 4 epsilon closure did not converge, 5 not decoded), tokens and links created, cost of
 * the best path.  Returns PK2_ERR_LIMIT / PK2_ERR_NUMERIC if any utterance failed.  Any output may be NULL. */
int pk2_lattice_summary(const pk2_lattice_batch* b, const void* workspace, int32_t* status,
                        int32_t* num_tokens, int32_t* num_links, float* best_cost, void* stream);
/* MMI: post[n][t][pdf] (+)= numerator - denominator posterior (lattice scaled by lm_scale / acoustic_scale
 * first, as fst::ScaleLattice does: float weights multiplied in double and stored back as float; 1.0 / 0.2
 * at reference ops/ops.py:58); frames whose reference transition-id is not in the lattice
 * are dropped when drop_frames != 0.  ref_tids: device i32, utterance n at n*ref_stride.  post rows must
 * be zero on entry.  lat_like: device f64[num_seq] = total log-likelihood of each lattice. */
int pk2_lattice_mmi(const pk2_lattice_batch* b, void* workspace, const int32_t* ref_tids,
                    int64_t ref_stride, const int32_t* tid2pdf, double lm_scale, double acoustic_scale,
                    int32_t drop_frames, float* post, int64_t post_seq_stride,
                    int64_t post_frame_stride, double* lat_like, void* stream);
/* sMBR (criterion 0) / MPFE (criterion 1): post (+)= arc posterior * (expected accuracy through the arc -
 * expected accuracy); score: device f64[num_seq] = expected frame accuracy.  tid2phone: device i32
 * [num_tids + 1]; phone_is_silence: device u8 [num_phones + 1]. */
int pk2_lattice_mpe(const pk2_lattice_batch* b, void* workspace, const int32_t* ref_tids,
                    int64_t ref_stride, const int32_t* tid2pdf, const int32_t* tid2phone,
                    const uint8_t* phone_is_silence, int32_t criterion, int32_t one_silence_class,
                    double lm_scale, double acoustic_scale, float* post, int64_t post_seq_stride,
                    int64_t post_frame_stride, double* score, void* stream);
/* Test / tooling hook: copies the pruned lattice of utterance n to host arrays (synchronises `stream`).
 * Call with null arrays to get the counts.  Tokens are renumbered frame by frame; link_ac is the acoustic
 * cost with the acoustic scale removed. */
int pk2_lattice_export(const pk2_lattice_batch* b, const void* workspace, int32_t n, int32_t* num_tokens,
                       int32_t* num_links, int32_t* tok_frame, int32_t* tok_state, float* tok_cost,
                       float* tok_final, int32_t* link_src, int32_t* link_dst, int32_t* link_tid,
                       float* link_graph, float* link_ac, void* stream);

/* Lattice determinisation on word labels (host arrays in, host handle out; no device work).  Replaces, for the lattice
 * dump of reference bin/latgen.py:143-181, what PyKaldi's recogniser does when `determinize_lattice = True`
 * (bin/latgen.py:149: Kaldi's DeterminizeLatticePhonePrunedWrapper): the result accepts every word sequence of the raw
 * state-level lattice once, with the (graph, acoustic) costs and the transition-id alignment of its best path
 * (CompactLattice semantics, no minimisation).  Arc i of the input: src[i] -> dst[i], word[i] (0 = epsilon), tid[i] (0 =
 * none), costs graph[i], acoustic[i]; final_cost[s] (+inf = not final).  beam: what lies on no complete path within beam
 * of the best one is dropped, before and after the subset construction (pass the decoder's lattice beam); more than
 * max_states output states -> PK2_ERR_LIMIT (retry with a smaller beam, as Kaldi's wrapper does). */
typedef struct pk2_det_lattice pk2_det_lattice;
int pk2_lattice_determinize(int32_t num_states, int32_t start, int64_t num_arcs, const int32_t* src, const int32_t* dst,
                            const int32_t* word, const int32_t* tid, const float* graph, const float* acoustic,
                            const float* final_cost, double beam, int64_t max_states, pk2_det_lattice** out);
int pk2_det_lattice_sizes(const pk2_det_lattice* h, int32_t* num_states, int32_t* start, int64_t* num_arcs,
                          int64_t* arc_tids, int64_t* final_tids);
/* Arcs sorted by (source state, word): src, dst, word, graph, acoustic [num_arcs], tid_off [num_arcs + 1] into tids;
 * per state final_graph / final_acoustic (+inf = not final) and final_tid_off [num_states + 1] into final_tids. */
int pk2_det_lattice_export(const pk2_det_lattice* h, int32_t* src, int32_t* dst, int32_t* word, float* graph,
                           float* acoustic, int64_t* tid_off, int32_t* tids, float* final_graph, float* final_acoustic,
                           int64_t* final_tid_off, int32_t* final_tids);
void pk2_det_lattice_destroy(pk2_det_lattice* h);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange: RCCL all-reduce over xGMI, one communicator per process (= per GPU).
 * Replaces horovod.torch's NCCL all-reduce hidden in hvd.DistributedOptimizer.step()
 * (reference bin/train_chain.py:141-145, bin/train_ce.py:127-131, bin/train_se.py:130-134).
 * librccl is bound at run time (dlopen; a copy already mapped into the process is reused; PK2_RCCL_LIB
 * overrides).  Only the unique id travels through the launcher's own channel (torch.distributed store /
 * broadcast in pykaldi2_amd/hvd.py, MPI in the reference).
 * ---------------------------------------------------------------------------------------------- */
typedef struct pk2_comm pk2_comm;
/* Size of a unique id in bytes (NCCL_UNIQUE_ID_BYTES = 128). */
int32_t pk2_comm_unique_id_bytes(void);
/* Rank 0: generates the id every rank must pass to pk2_comm_init (host buffer of pk2_comm_unique_id_bytes()). */
int pk2_comm_unique_id(void* id_out);
/* Collective over all ranks (ncclCommInitRank on the calling thread's current device). */
int pk2_comm_init(int32_t rank, int32_t world, const void* unique_id, pk2_comm** out);
/* In-place SUM all-reduce of buf[0..count) (device f32: a bucket of the flat gradient buffer) enqueued on `stream`
 * (hipStream_t): the compute stream, or a side stream the caller fences with events.  Averaging (1/world) is folded
 * into the optimiser kernel (pk2_adam_step / pk2_sgd_step grad_scale), not done here. */
int pk2_allreduce_bucket(pk2_comm* comm, float* buf, int64_t count, void* stream);
/* The same plus, in the same RCCL group (one launch), the max over the ranks of the one-float guard slot
 * (pk2_persist_guard_export / _import above). */
int pk2_allreduce_guarded(pk2_comm* comm, float* buf, int64_t count, float* guard_slot, void* stream);
/* rank / world of the communicator and the path of the RCCL library in use (any output may be NULL). */
int pk2_comm_info(const pk2_comm* comm, int32_t* rank, int32_t* world, char* lib_path, int32_t lib_path_bytes);
int pk2_comm_destroy(pk2_comm* comm);

/* ------------------------------------------------------------------------------------------------
 * Side streams confined to a part of the chip (no reference counterpart: cuBLAS / cuDNN own the whole GPU).  The weight
 * gradients of a layer hang off the serial chain of the backward pass (reference models/lstm.py:49-58 through autograd);
 * pykaldi2_amd.lstm can launch them on a side stream (PK2_SIDE_STREAM=1), and with PK2_SIDE_CU_PER_XCD=k that stream is
 * created with a CU mask: the first k CUs of every XCD (hipExtStreamCreateWithCUMask; mask bit i = CU i / 8 of XCD i % 8 on
 * this 8-XCD part -- pk2_debug_where reports where the workgroups of a launch really ran).
 * ---------------------------------------------------------------------------------------------- */
int pk2_stream_create_cu_mask(int32_t cus_per_xcd, void** stream_out);
int pk2_stream_destroy(void* stream);
/* out[3 * b + {0, 1, 2}] = XCC id, shader engine, CU of workgroup b of a `blocks`-workgroup launch on `stream` (HW_ID). */
int pk2_debug_where(int32_t* out, int32_t blocks, void* stream);
/* Stand-in for the collective library's all-reduce kernel on a one-GPU box (PK2_HVD_FAKE_PEER in pykaldi2_amd/hvd.py):
 * `blocks` workgroups stream dst[i] += src[i] over n floats, `passes` times, on `stream`. */
int pk2_debug_peer_reduce(float* dst, const float* src, int64_t n, int32_t blocks, int32_t passes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PK2HIP_H_ */
