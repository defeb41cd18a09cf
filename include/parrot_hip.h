/*
 * parrot_hip.h -- C ABI of libparrot_hip.so: the MI355X (gfx950) implementation of the
 * Parrot-TTS synthesis hot path (TTE -> length regulator -> HiFi-GAN unit vocoder).
 *
 * The reference (parrot-tts/Parrot-TTS) is pure Python on torch.nn and has no FFI of its own;
 * the seam this library plugs into is the nn.Module surface the reference drivers call.  Each
 * entry point below names the reference code it replaces (paths relative to the reference
 * repo).  The Python shims in parrot_tts_amd/ (tte.py, vocoder.py) bind these with ctypes --
 * see INTEGRATION.md.  Test / profiling / probe entry points live in parrot_hip_debug.h.
 *
 * Conventions
 *   - plain C types only; no torch / C++ types cross the boundary
 *   - return 0 on success, a negative PARROT_E_* code otherwise; parrot_last_error() gives a
 *     thread-local message; no C++ exception crosses the ABI
 *   - weights are HOST pointers (fp32, row-major, already weight-norm-folded): *_create packs them
 *     into MFMA fragment order and uploads them once; the handle owns that device copy
 *   - activations / ids / outputs are DEVICE pointers allocated by the caller (PyTorch's caching
 *     allocator); the library never allocates per call: callers pass a workspace sized by
 *     *_workspace_bytes().  All launches are asynchronous on the hipStream_t passed (as void*)
 *   - one process per GPU; a handle is bound to the device current at *_create
 */
#ifndef PARROT_HIP_H
#define PARROT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARROT_ABI_VERSION 7  /* 7: parrot_voc_wait_stage */

enum {
    PARROT_OK = 0,
    PARROT_E_INVALID = -1,     /* bad argument / unsupported dimension                         */
    PARROT_E_RANGE = -2,       /* T >= max_len (reference: IndexError at modules/fft.py:18)    */
    PARROT_E_HIP = -3,         /* a HIP runtime call failed                                    */
    PARROT_E_NOMEM = -4,       /* workspace too small / allocation failed                      */
    PARROT_E_UNSUPPORTED = -5, /* configuration outside what the kernels cover                 */
    PARROT_E_NONFINITE = -6,   /* parrot_voc_check: a waveform sample was NaN / inf (fp16 split range exceeded) */
};

int parrot_abi_version(void);
const char* parrot_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Single Conv1d / ConvTranspose1d plan: the dilated-Conv1d implicit-GEMM kernel behind every
 * torch.nn.Conv1d / ConvTranspose1d / Linear on the path
 *   (utils/vocoder/models.py:17-28,75,81-83,91; modules/fft.py:48-50,65-76; modules/duration.py:64-72).
 * y[b,o,t] = act(bias[o] + sum_{i,j} w[o,i,j] * pre(x[b,i,t + j*dil - pad])) (+ res[b,o,t])
 * ------------------------------------------------------------------------------------------ */
typedef struct parrot_conv parrot_conv_t;

typedef struct {
    int32_t c_in, c_out, k, dilation, padding; /* Conv1d: symmetric zero padding                 */
    int32_t transposed;                        /* 1: ConvTranspose1d, weight (c_in,c_out,k)      */
    int32_t stride;                            /* ConvTranspose1d stride (upsample rate); else 1 */
    int32_t pre_act;                           /* 0 none, 1 leaky_relu(pre_slope) on the input   */
    float pre_slope;
    int32_t act;                               /* 0 none, 1 relu, 2 tanh (applied to bias+sum)   */
    int32_t tile_cfg;                          /* -1 = auto; else index into the tile table      */
    int32_t precision;                         /* PARROT_PREC_*; -1 = library default            */
} parrot_conv_desc;

/* How the fp32 products of layers with >= 16 channels are evaluated.  Inputs, outputs, residual stream and accumulation
 * are fp32 in every mode (csrc/conv_split.h has the details).
 *   PARROT_PREC_F32    v_mfma_f32_32x32x2_f32: exact fp32 fma chain.
 *   PARROT_PREC_F16X3  (default) each operand split into 2 fp16 pieces (pre-scaled by powers of two), 3 fp16 MFMAs per
 *                      product group (dropped term <= 2^-22 of the product), fp32 accumulate: fp32-class error, held to
 *                      the same parity tolerances as PARROT_PREC_F32, at 16/3 the MFMA rate.  Activations must stay
 *                      below 8190 in magnitude (fp16 range after the 2^3 pre-scale): beyond that the output turns
 *                      into inf/NaN, never into a silently wrong finite value.
 *   PARROT_PREC_BF16X6 each operand split into 3 bf16 pieces, 6 bf16 MFMAs per product group (dropped terms <= 2^-23),
 *                      fp32's full exponent range; same tolerances, 16/6 the MFMA rate.
 *   PARROT_PREC_BF16 / PARROT_PREC_F16   operands rounded once to bf16 / fp16, ONE MFMA per product group, fp32
 *                      accumulate, fp32 residual stream: the reduced-precision operating point (BASELINE configs[2]
 *                      "bf16"); NOT parity-grade -- reported by SNR against the fp32 result (36 dB / 54 dB).
 * The process-wide default (PARROT_PRECISION env: "f32" | "f16x3" | "bf16x6" | "bf16" | "f16", else f16x3)
 * is read ONCE by every *_create: a handle keeps the mode it was created under and is immutable afterwards. */
#define PARROT_PREC_F32 0
#define PARROT_PREC_BF16X6 1
#define PARROT_PREC_F16X3 2
#define PARROT_PREC_BF16 3
#define PARROT_PREC_F16 4
/* (Per handle: parrot_voc_create_ex / parrot_tte_create_ex / parrot_conv_desc.precision.  The process-wide DEFAULTS -- precision,
 * fused ResBlock kernels (env PARROT_FUSED: 0 off, 1 every eligible stage, 2 default), merged TTE projections (env PARROT_TTE_MERGE) --
 * come from the environment; their setters are test / bench conveniences and live in parrot_hip_debug.h.) */

int parrot_conv_create(parrot_conv_t** out, const parrot_conv_desc* d, const float* w_host, const float* bias_host);
void parrot_conv_destroy(parrot_conv_t*);
/* epilogue: 0 store, 1 y += v, 2 y = (y + v) / div  (MRF sum, models.py:100-106) */
int parrot_conv_run(parrot_conv_t*, const float* x, const float* res, float* y, int32_t B, int32_t T_in,
                    int32_t epilogue, float div, void* stream);
int parrot_conv_out_len(const parrot_conv_t*, int32_t T_in);

/* ------------------------------------------------------------------------------------------
 * HiFi-GAN unit vocoder: CodeGenerator.forward (utils/vocoder/models.py:153-169) ->
 * Generator.forward (:95-111) with ResBlock1/2 (:31-38,:58-62).
 * ------------------------------------------------------------------------------------------ */
typedef struct parrot_voc parrot_voc_t;

#define PARROT_MAX_STAGES 8
#define PARROT_MAX_KERNELS 4
#define PARROT_MAX_DIL 4

typedef struct {
    int32_t num_embeddings, embedding_dim; /* h.num_embeddings, h.embedding_dim                 */
    int32_t multispkr, n_spkr;             /* bool(h.multispkr); spkr table rows (10)           */
    int32_t model_in_dim;                  /* conv_pre input channels                           */
    int32_t upsample_initial_channel;
    int32_t n_stages;
    int32_t upsample_rates[PARROT_MAX_STAGES];
    int32_t upsample_kernel_sizes[PARROT_MAX_STAGES];
    int32_t n_kernels;
    int32_t resblock_kernel_sizes[PARROT_MAX_KERNELS];
    int32_t n_dil;
    int32_t resblock_dilation_sizes[PARROT_MAX_KERNELS][PARROT_MAX_DIL];
    int32_t resblock_type;                 /* 1 = ResBlock1, 2 = ResBlock2 (h.resblock)         */
} parrot_voc_cfg;

/* Folded fp32 weights on the HOST, torch layouts.  resblock conv index:
 *   ResBlock1: [(stage*n_kernels + j)*2*n_dil + m*2 + {0: convs1[m], 1: convs2[m]}]
 *   ResBlock2: [(stage*n_kernels + j)*n_dil + m]                                              */
typedef struct {
    const float* dict;      /* (num_embeddings, embedding_dim)  */
    const float* spkr;      /* (n_spkr, embedding_dim) or NULL  */
    const float* conv_pre_w; const float* conv_pre_b;   /* (C0, model_in_dim, 7), (C0)          */
    const float* ups_w[PARROT_MAX_STAGES];               /* (C_in, C_in/2, k)                    */
    const float* ups_b[PARROT_MAX_STAGES];
    const float* const* rb_w;                            /* array of (C, C, k) pointers          */
    const float* const* rb_b;
    int32_t n_rb;
    const float* conv_post_w; const float* conv_post_b; /* (1, C_last, 7), (1)                  */
} parrot_voc_weights;

int parrot_voc_create(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w);
/* The same with this handle's own precision (PARROT_PREC_*, -1 = the process default) and fused-ResBlock mode (0 / 1 / 2,
 * -1 = default): what the shims' range-safe fallback uses to rebuild a handle in PARROT_PREC_BF16X6 (fp32's range) after the
 * default fp16x3 scheme reported PARROT_E_NONFINITE -- reference checkpoints carry no range promise (utils/vocoder/models.py has
 * no normalisation layer), so |activation| < 8190 cannot be assumed of a real one. */
int parrot_voc_create_ex(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w, int32_t precision,
                         int32_t fused_resblocks);
void parrot_voc_destroy(parrot_voc_t*);
/* PARROT_PREC_* this handle was created with (the "precision in use" after a fallback). */
int parrot_voc_precision(const parrot_voc_t*);
size_t parrot_voc_workspace_bytes(const parrot_voc_t*, int32_t B, int32_t U);
/* code (B,U) int64, spkr (B,1) int64 or NULL -> wav (B,1,U*prod(rates)) fp32 in (-1,1).
 * unit_lens: optional (B) int32 device array of real units per row (ragged batch padded to U): every layer applies its
 * zero padding at each row's own end, so row b[: unit_lens[b]*hop] equals the reference's B=1 run of that utterance
 * (the reference never batches the vocoder, utils/vocoder/inference.py:149); samples beyond are unspecified.  NULL = all U.
 * stage_out: optional array of 2*n_stages+1 device pointers (conv_pre, ups_i, mrf_i ...) that
 * receive copies of the intermediate activations (tests only); NULL in production.            */
int parrot_voc_forward(parrot_voc_t*, const int64_t* code, const int64_t* spkr, const int32_t* unit_lens, int32_t B, int32_t U,
                       float* wav_out, float* const* stage_out, void* ws, size_t ws_bytes, void* stream);
/* CodeGenerator.forward with extra conditioning keywords (utils/vocoder/models.py:162-167: every keyword tensor other
 * than code / spkr / f0 is upsampled to U frames and concatenated behind the embeddings).  `feats`: dense fp32
 * (B, n_feat_channels, U), the caller's streams already upsampled and concatenated in keyword order; it fills input
 * channels [embedding_dim * (1 + multispkr), model_in_dim).  NULL / 0 when the model has none. */
int parrot_voc_forward_feats(parrot_voc_t*, const int64_t* code, const int64_t* spkr, const float* feats, int32_t n_feat_channels,
                             const int32_t* unit_lens, int32_t B, int32_t U, float* wav_out, float* const* stage_out, void* ws,
                             size_t ws_bytes, void* stream);
/* Chunk-streamed synthesis (long-form utterances, BASELINE configs[4]): chunks of `chunk_units` units are vocoded with
 * `halo_units` units of real context on both sides (< 0: the generator's receptive field, parrot_voc_receptive_units: 21 units for the shipped config) and only
 * their own samples land in wav_out (B,1,U*hop): equal to parrot_voc_forward on the whole utterance to fp32 round-off, with
 * the activation memory of chunk_units + 2*halo_units units.  Not for models with extra conditioning streams. */
size_t parrot_voc_chunked_workspace_bytes(const parrot_voc_t*, int32_t B, int32_t chunk_units, int32_t halo_units);
int parrot_voc_forward_chunked(parrot_voc_t*, const int64_t* code, const int64_t* spkr, const int32_t* unit_lens, int32_t B, int32_t U,
                               int32_t chunk_units, int32_t halo_units, float* wav_out, void* ws, size_t ws_bytes, void* stream);
/* Synchronises `stream` and reports what the device flagged since the last check: PARROT_E_RANGE for a unit / speaker id
 * outside the embedding tables (the reference's IndexError), PARROT_E_NONFINITE when a waveform sample left [-1, 1] as NaN /
 * inf (an activation beyond the fp16 split scheme's range: never a silently wrong finite value). */
int parrot_voc_check(parrot_voc_t*, void* stream);
/* The same flag WITHOUT a synchronisation: copies its value (0 = ok, 1 / 2 bad unit / speaker id, 5 non-finite sample) to
 * dst_dev[0] (device memory) on `stream` and clears it -- for callers that fetch it with a device-to-host copy they do anyway. */
int parrot_voc_status_async(parrot_voc_t*, int32_t* dst_dev, void* stream);
/* ... and without clearing it: the shims look at it once, synchronously, after the FIRST forward of every handle (range-safe
 * fallback: 5 -> rebuild in PARROT_PREC_BF16X6 and re-run); any other value stays set for the regular reporting path. */
int parrot_voc_status_peek_async(parrot_voc_t*, int32_t* dst_dev, void* stream);
/* Make `stream` wait until the most recently enqueued (direct, non-graph) forward of this handle has reached MRF stage `stage`
 * (0 .. n_stages - 1): a caller that runs other work beside the forward -- SynthesisPipeline.submit: the next batch's TTE, reference
 * inference.py + utils/vocoder/inference.py back to back -- chooses which part of the forward it shares the chip with.  Before the
 * handle's first forward (and after a graph replay of a small shape) there is nothing to wait for: the call returns at once. */
int parrot_voc_wait_stage(parrot_voc_t*, int32_t stage, void* stream);
/* Receptive field of the generator in units, either side of an output frame, from the handle's configuration (interval
 * propagation through conv_post, the MRF stages, the transposed convs of reference utils/vocoder/models.py:80-83 and conv_pre):
 * 21 for the shipped config.  The default halo of parrot_voc_forward_chunked. */
int parrot_voc_receptive_units(const parrot_voc_t*);
/* Waveform samples of an utterance of U units: U * prod(upsample_rates) for the shipped configs; a stage with odd
 * upsample_kernel_size - upsample_rate adds one sample (ConvTranspose1d then yields T u + 1, models.py:80-83): wav_out of
 * parrot_voc_forward holds B rows of parrot_voc_out_len(U) samples, and a row of n units has parrot_voc_out_len(n) real ones. */
int64_t parrot_voc_out_len(const parrot_voc_t*, int32_t U);
/* wav (n) fp32 -> int16 as `(x*32768).astype('int16')` does (utils/vocoder/inference.py:71-73) */
int parrot_wav_to_int16(const float* wav, int16_t* out, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * TTE: Parrot.forward(inference=True) / Parrot.infer  (modules/parrot.py:90-120) with
 * FFTBlock (modules/fft.py:85-100), DurationPredictor + length_regulator (modules/duration.py),
 * get_mask_from_lengths (modules/data.py:8-20).
 * ------------------------------------------------------------------------------------------ */
typedef struct parrot_tte parrot_tte_t;

typedef struct {
    int32_t d_model, n_filter_ffn, ffn_k1, ffn_k2, max_len;
    int32_t enc_layers, enc_heads, dec_layers, dec_heads;
    int32_t dp_filter, dp_kernel;
    int32_t vocab, n_speaker /* 0/1 = no speaker_emb */, n_codes /* head width (hubert_codes) */;
} parrot_tte_cfg;

typedef struct {
    const float *qkv, *in_proj, *out_proj, *wo;                   /* (3D,D) (3D,D) (D,D) (D,D)  */
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b;           /* (F,D,k1) (F) (D,F,k2) (D)  */
    const float *attn_norm_w, *attn_norm_b, *conv_norm_w, *conv_norm_b;
} parrot_fft_weights;

typedef struct {
    const float* pe;         /* (max_len, D)  pos_emb.pe                                        */
    const float* tok_emb;    /* (vocab, D)                                                       */
    const float* speaker_emb;/* (n_speaker, D) or NULL                                           */
    const float *dp_conv0_w, *dp_conv0_b, *dp_ln0_w, *dp_ln0_b;
    const float *dp_conv1_w, *dp_conv1_b, *dp_ln1_w, *dp_ln1_b;
    const float *dp_proj_w, *dp_proj_b;
    const parrot_fft_weights* enc;  /* enc_layers entries */
    const parrot_fft_weights* dec;  /* dec_layers entries */
    const float *head_w, *head_b;   /* (n_codes, D), (n_codes) */
} parrot_tte_weights;

int parrot_tte_create(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w);
/* The same with this handle's own precision (PARROT_PREC_*, -1 = default) and projection merge (0 / 1, -1 = default): the
 * range-safe fallback of the shims, and the merged-vs-unmerged rows of the parity report. */
int parrot_tte_create_ex(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w, int32_t precision,
                         int32_t merge_projections);
void parrot_tte_destroy(parrot_tte_t*);
int parrot_tte_precision(const parrot_tte_t*);
/* `state` carries the encoder output + duration prefix sums from encode to decode (sized by B,S);
 * `ws` is scratch: encode needs workspace_bytes(B,S,0), decode workspace_bytes(B,S,L).         */
size_t parrot_tte_state_bytes(const parrot_tte_t*, int32_t B, int32_t S);
size_t parrot_tte_workspace_bytes(const parrot_tte_t*, int32_t B, int32_t S, int32_t L_max);
/* Phase 1 (parrot.py:94-102 up to the durations): phones (B,S) i64, src_mask (B,S) u8 1=valid,
 * speaker (B) i64 or NULL -> log_dur (B,S) f32, dur (B,S) i64, out_lens (B) i32 (sum of dur).
 * src_len: NULL = the reference's PADDED-BATCH semantics (Parrot.forward on the padded batch: pe[S] of the padded length,
 *   pad frames leak through the k = 9 / k = 3 convs -- quirk Q7: a row's result depends on the batch it is padded into).
 *   (B) i32 device = ROW-EXACT mode: row b holds src_len[b] real tokens (src_mask must be that prefix) and is evaluated as the
 *   reference evaluates that utterance ALONE (its drivers run batch_size = 1, inference.py:34): pe[src_len[b]] (fft.py:18), every
 *   conv zero-padded at the row's own end (fft.py:78-82, duration.py:64-72), keys beyond it masked.                          */
int parrot_tte_encode(parrot_tte_t*, const int64_t* phones, const uint8_t* src_mask, const int64_t* speaker,
                      const int32_t* src_len /* nullable */, int32_t B, int32_t S, float* log_dur, int64_t* dur, int32_t* out_lens,
                      void* state, size_t state_bytes, void* ws, size_t ws_bytes, void* stream);
/* Phase 2 (duration.py:6-24, parrot.py:106-108,115): needs L = max(out_lens) from the host
 * (the reference's own host sync, duration.py:10).  -> ids (B,L) i64 argmax, tgt_mask (B,L) u8
 * (ids <= len, quirk Q2), optional logits (B,L,n_codes) f32 (tests).
 * row_exact != 0 (after an encode with src_len): row b is decoded as its own B = 1 run -- pe[out_lens[b]] (parrot.py:106), convs
 * zero-padded and keys masked at out_lens[b]; tgt_mask[b] = t < max(out_lens[b], 1): the row's ids are ids[b, :out_lens[b]]
 * (alone, a row is the longest of its batch and emits exactly its length -- no extra frame).                                */
int parrot_tte_decode(parrot_tte_t*, int32_t B, int32_t S, int32_t L, int32_t row_exact,
                      int64_t* ids, uint8_t* tgt_mask, float* logits /* nullable */,
                      void* state, size_t state_bytes, void* ws, size_t ws_bytes, void* stream);
/* Device-side flags.  Synchronises `stream`, clears the flag; returns 0, PARROT_E_RANGE (a bad phone / speaker id <-> the
 * reference's Embedding IndexError) or PARROT_E_NONFINITE (NaN / inf logits at some position of the last decode: an
 * activation beyond the fp16 split scheme's range -- the ids of that call are not to be trusted). */
int parrot_tte_check(parrot_tte_t*, void* stream);
/* Tie guard of the unit-id argmax (reference modules/parrot.py:115: torch.argmax over 1000 fp32 logits).  parrot_tte_decode
 * measures every position's top-2 margin; where it is below PARROT_TIE_GUARD (default 1e-4; 0 = off) the head is re-evaluated
 * for that position in fp64 (exact products of the fp32 weights and activations, no accumulation-order dependence) and the
 * argmax taken again.  dst_dev[0..2] <- {guarded positions of the last decode, ids the re-evaluation changed, its smallest
 * margin (float bits)}; no synchronisation.  What no implementation can do is follow the reference below ITS OWN noise: its
 * fp32 logits move by ~1.2e-5 with the CPU thread count (tests/test_oracle_golden.py), so an id with a smaller margin is not
 * determined by the reference itself. */
int parrot_tte_guard_stats_async(parrot_tte_t*, int32_t* dst_dev, void* stream);
/* The guard's re-evaluation starts one layer before the head when the last decoder block's conv2 is 1x1 (the shipped config): that
 * conv2 + bias + residual (modules/fft.py:81,99) are recomputed in fp64 from the block's own fp32 intermediates, then the head.
 * logits_dev (max_n x n_codes floats) / list_dev (2 max_n ints: (b, t) pairs) <- the refined logits of the guarded positions of
 * the last decode (first guard_stats[0], at most 256); device memory, no synchronisation.  Tests / parity reports. */
int parrot_tte_guard_logits(parrot_tte_t*, float* logits_dev, int32_t* list_dev, int32_t max_n, void* stream);
/* The same flag without a synchronisation (3 / 4 bad phone / speaker id, 5 non-finite logits): see parrot_voc_status_async. */
int parrot_tte_status_async(parrot_tte_t*, int32_t* dst_dev, void* stream);
int parrot_tte_status_peek_async(parrot_tte_t*, int32_t* dst_dev, void* stream);
/* length_regulator alone (modules/duration.py:6-24 + modules/data.py:8-20), on the kernel the decoder uses:
 * seq (B,S,D) f32 as the reference holds it, dur (B,S) i64 -> out (B,L,D) (rows repeat_interleave'd, zero right-padded),
 * mask (B,L) u8 = ids <= len (quirk Q2), out_lens (B) i32.  L = max over rows of sum(dur), computed by the caller like
 * duration.py:10; ws: 2*B*D*max(S,L) floats + B*S + B int32 (parrot_length_regulator_workspace_bytes). */
size_t parrot_length_regulator_workspace_bytes(int32_t B, int32_t S, int32_t D, int32_t L);
int parrot_length_regulator(const float* seq, const int64_t* dur, int32_t B, int32_t S, int32_t D, int32_t L, float* out,
                            uint8_t* mask, int32_t* out_lens, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARROT_HIP_H */
