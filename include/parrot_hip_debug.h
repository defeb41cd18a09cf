/*
 * parrot_hip_debug.h -- test, profiling and probe entry points of libparrot_hip.so.  NOT part of the interface a user switching
 * from the reference needs (that is parrot_hip.h); bench.py, the tests and the tools under tools/ bind these.
 */
#ifndef PARROT_HIP_DEBUG_H
#define PARROT_HIP_DEBUG_H

#include "parrot_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Runs the MFMA fragment-layout probe on the current device (0 = layouts as assumed). */
int parrot_selftest(void* stream);
/* Rows of the per-kernel timing table (tile configurations of the exact-fp32 kernel; the other kernel rows follow). */
int parrot_conv_num_tile_cfgs(void);

/* dst[i] = src[i], 4 bytes per lane: known-byte-count kernel for calibrating the HBM PMC counters. */
/* Process-wide DEFAULTS for handles created afterwards (tests and benches that build several handles under different modes in one
 * process; a product passes the mode per handle to *_create_ex).  Read once by every *_create; a live handle never changes. */
int parrot_set_default_precision(int32_t prec);
/* Fused ResBlock kernels (csrc/resblock_split.h, resblock_fused.h): 0 off (layer by layer), 1 every eligible stage,
 * 2 (default; env PARROT_FUSED) all but the exact-fp32 32-channel kernel.  Default for handles created afterwards. */
int parrot_set_fused_resblocks(int32_t mode);
/* FFT blocks project twice on each side of the attention core (quirk Q3, modules/fft.py:48-57: qkv then MHA in_proj; MHA
 * out_proj then wo; all bias-free).  1 (default; env PARROT_TTE_MERGE): each pair is evaluated as its fp64-formed product, one
 * launch; 0: one after the other as the reference does.  Default for handles created afterwards. */
int parrot_set_tte_merge(int32_t on);
int parrot_debug_copy(const float* src, float* dst, size_t n, void* stream);
/* Measurement aid (bench.py `roofline.ceiling_probe_tflops`): the rate a bare fp16 MFMA stream sustains on this device under
 * its power limit -- shape 0 = v_mfma_f32_32x32x16_f16, 1 = v_mfma_f32_16x16x32_f16; random operands (constant_data = 0) or one
 * constant (1); two waves per SIMD, ~20-40 ms.  Synchronises the device.  TFLOP/s of 16-bit MFMA work in *tflops_out. */
int parrot_debug_mfma_ceiling(int32_t shape, int32_t constant_data, double* tflops_out);
/* Per-launch timing of the conv kernels (HIP events on the launch stream, aggregated per kernel row: bench.py's roofline
 * object).  parrot_prof_begin times every launch; parrot_prof_begin_row only the launches of one row (the dominant kernel):
 * event records around every launch of a step are themselves 3 % of a B = 64 step and 20 % of a single-utterance one.
 * parrot_prof_end: out[4 row + {0,1,2,3}] = {launches, total ms, algorithmic flops, algorithmic bytes}. */
int parrot_prof_begin(void);
int parrot_prof_begin_row(int32_t row);
int parrot_prof_end(double* out, int32_t n_cfg);

/* Debug aid: headroom to the fp16 split scheme's range.  While dst_dev != NULL every conv launched by parrot_voc_forward records
 * max |input element| into dst_dev[group] (device floats, atomic max; the caller zeroes them): group 0 = conv_pre, 1 + i = the
 * layers of stage i (ups_i, its ResBlock convs), n_stages + 1 = conv_post.  Fused ResBlock launches only see their block's input:
 * create the handle with fused_resblocks = 0 to cover every layer.  NULL switches it off. */
int parrot_voc_debug_absmax(parrot_voc_t*, float* dst_dev);
/* Tests / error localisation: while set, the next encode / decode calls copy the channel-first (B, D, T) activation
 * after each stage to the given DEVICE buffers (NULL entries are skipped): enc_ptrs[0] = embedding + pe[S],
 * enc_ptrs[1 + n] = encoder block n, enc_ptrs[1 + enc_layers] = encoder output (+ speaker); dec_ptrs[0] = length
 * regulator output + pe[L], dec_ptrs[1 + n] = decoder block n.  Pass NULL, NULL to switch it off. */
int parrot_tte_debug_stages(parrot_tte_t*, float* const* enc_ptrs, float* const* dec_ptrs);

#ifdef __cplusplus
}
#endif
#endif /* PARROT_HIP_DEBUG_H */
