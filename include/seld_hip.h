/* libseld_hip.so -- C ABI of the MI355X (gfx950) SELD hot path.
 *
 * The reference (Zeudon/sound-event-localization-detection) is pure Python on stock PyTorch /
 * torchaudio and has NO FFI of its own; its "plugin surface" is the set of Python names main.py
 * imports.  This header is the boundary a maintainer binds with ctypes
 * (INTEGRATION.md shows the stub): each entry point names the reference call it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure
 *     (-1 invalid argument, -2 HIP runtime error, -3 seld_init not called, -4 unsupported);
 *     seld_last_error() returns the text of the calling thread's last failure.
 *   - all data pointers are CALLER-OWNED DEVICE pointers (e.g. torch.Tensor.data_ptr());
 *     the library never allocates outputs and never frees inputs.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is asynchronous on it, the library never synchronises.
 *   - int64_t extents; tensors are dense row-major in the layout written next to each pointer.
 *   - one caller thread per device (one process per GPU under torchrun).
 */
#ifndef SELD_HIP_H_
#define SELD_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library state -------------------------------------------------------------------- */
int seld_init(int device);            /* builds the constant tables (Hann, twiddles, sparse mel, GCC cosine / sine) */
int seld_shutdown(void);
int seld_version(void);
const char* seld_last_error(void);

/* Override the default HTK mel filterbank with a host table fb[481][64] (fp32), e.g. the one
 * torchaudio.functional.melscale_fbanks would build.  Replaces the constant inside
 * torchaudio.transforms.MelScale used at dataset.py:38-43. */
int seld_set_mel_filterbank(const float* fb_host);

/* Override the default (double-precision-rounded) periodic Hann window with a host table w[960],
 * e.g. torch.hann_window(960) as torchaudio.transforms.Spectrogram builds it in fp32. */
int seld_set_window(const float* window_host);

/* Host copies of the default tables -- no GPU needed (used by CPU tests). Any pointer may be NULL. */
int seld_default_tables(float* window960, float* fb481x64, int* mel_b0_64, float* mel_wd_24x64,
                        float* mel_wu_24x64);

/* ---- features: dataset.py:27-58 audio_to_mel_spectrogram ------------------------------ */
/* Number of STFT frames for L samples (center=True, hop 480): 1 + L/480. */
int64_t seld_num_frames(int64_t L);

/* Fused reflect-pad -> frame(960, hop 480) -> periodic Hann -> 960-pt rFFT -> |X|^2 -> HTK mel(64)
 * -> 10*log10(max(.,1e-10)).   pcm [N][C][L] (float in [-1,1), or int16 = float*32768),
 * F = seld_num_frames(L).
 *   layout 0: out [N][C][64][F]   (the reference's [C, n_mels, T] per clip, dataset.py:53)
 *   layout 1: out [N][F][C][64]   (time-major: a training window is a contiguous slice,
 *                                  i.e. the permute(2,0,1) of dataset.py:303 is free)
 * Requires L > 480 (reflect padding), like torch.stft. */
int seld_logmel_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, void* stream);
int seld_logmel_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, void* stream);

/* Same kernels with caller-defined output strides (elements): out[n*sN + c*sC + m*sM + t*sT] -- used to write
 * the log-mel channels into a wider time-major feature tensor [N][F][C_total][64] next to the spatial features. */
int seld_logmel_f32_strided(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, void* stream);
int seld_logmel_i16_strided(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, void* stream);

/* The strided log-mel pass that ALSO writes the spectra it squares, spec_complex [N][C][F][481] complex64 (what
 * seld_stft_* returns): the spatial features below read them, and one pass over the PCM serves both (north-star
 * additions A14-A16; the reference has no such call -- its STFT lives inside torchaudio, dataset.py:27-58). */
int seld_logmel_spectrum_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                             int64_t sM, int64_t sT, float* spec_complex, void* stream);
int seld_logmel_spectrum_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                             int64_t sM, int64_t sT, float* spec_complex, void* stream);

/* The strided log-mel pass that also writes every bin's PHASOR X / |X| as a pair of signed 16-bit fixed-point numbers
 * (word = (re & 0xffff) | im << 16, scale 32767; 0 for a silent bin, |X|^2 <= 1e-12): phasors_q15
 * [N][C][F][seld_phasor_pitch()] words, bins 0..480 of a row written.  This is what GCC-PHAT consumes (seld_gcc_phat_q15):
 * 4 B per bin instead of the 8 B of the complex64 spectrum -- half the HBM bytes between the two kernels -- with the
 * phase transform's reciprocal square root taken once per (channel, bin) where |X|^2 is formed anyway (extends
 * dataset.py:27-58; no upstream counterpart). */
int64_t seld_phasor_pitch(void);
int seld_logmel_phasors_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, uint32_t* phasors_q15, void* stream);
int seld_logmel_phasors_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, uint32_t* phasors_q15, void* stream);

/* ---- north-star additions without a reference implementation --------------
 * The reference computes its STFT only implicitly inside torchaudio and has no intensity-vector / GCC-PHAT
 * features; these entry points follow the DCASE SELD-baseline definitions.
 *
 * STFT (the complex spectrum the log-mel kernel squares): out_complex [N][C][F][481] complex64 (interleaved
 * re, im; frame-major).  torch.stft(...) layout is its transpose(-1, -2). */
int seld_stft_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out_complex, void* stream);
int seld_stft_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out_complex, void* stream);

/* FOA intensity vectors from spectra [N][4][F][481] (channel 0 = W):
 *   I_c[k] = Re(conj(W) X_c) / (1e-8 + |W|^2 + (|X_1|^2+|X_2|^2+|X_3|^2)/3),  iv[c][m] = sum_k fb[k][m] I_c[k]
 * out[n*sN + c*sC + m*sM + t*sT], c = 0..2, m = 0..63. */
int seld_foa_intensity(const float* spec_complex, int64_t N, int64_t F, float* out, int64_t sN, int64_t sC,
                       int64_t sM, int64_t sT, void* stream);

/* The FOA feature set in ONE pass over the PCM (pcm [N][4][L], channel 0 = W): log-mel of the four channels into out
 * channels 0..3 and the three mel-projected intensity vectors (same definition as seld_foa_intensity) into channels 4..6 of
 * out[n*sN + c*sC + m*sM + t*sT]; the spectra stay in LDS.  Extends dataset.py:27-58. */
int seld_logmel_iv_f32(const float* pcm, int64_t N, int64_t L, float* out, int64_t sN, int64_t sC, int64_t sM, int64_t sT,
                       void* stream);
int seld_logmel_iv_i16(const int16_t* pcm, int64_t N, int64_t L, float* out, int64_t sN, int64_t sC, int64_t sM, int64_t sT,
                       void* stream);

/* GCC-PHAT of all C(C-1)/2 channel pairs (m < n, lexicographic) from spectra [N][C][F][481], 2 <= C <= 8:
 *   cc = irfft(R/|R|, 960) with R = conj(X_m) X_n (factor 1 where either channel's bin is silent, |X|^2 <= 1e-12);
 *   out[n*sN + pair*sC + j*sM + t*sT] = cc[(j - 32) mod 960], j = 0..63 (lags -32..31).  With sM == 1 and 16-byte
 *   aligned rows the lags come from the matrix cores (fp16 operands, fp32 accumulation, <= 1e-4 abs vs float64);
 *   any other layout takes the fp32 FFT kernel. */
int seld_gcc_phat(const float* spec_complex, int64_t N, int64_t C, int64_t F, float* out, int64_t sN, int64_t sC,
                  int64_t sM, int64_t sT, void* stream);

/* The same lags from the Q15 phasors of seld_logmel_phasors_* (matrix-core kernel only: unit lag stride, 16-byte aligned
 * rows).  out as for seld_gcc_phat. */
int seld_gcc_phat_q15(const uint32_t* phasors_q15, int64_t N, int64_t C, int64_t F, float* out, int64_t sN, int64_t sC,
                      int64_t sM, int64_t sT, void* stream);

/* Host copy of the matrix-core kernel's constant operand -- no GPU needed (used by the CPU tests): 2 x 3 x 16 fragments of
 * 64 lanes x 8 IEEE binary16 values = 49 152 halves; fragment (part, tile, kstep), lane l, element j holds, for bin
 * k = 32 kstep + 8 (l >> 4) + j and lag n = 16 tile + (l & 15):  part 0: w_k cos(2 pi k n / 960),  part 1: -w_k sin(...),
 * w_0 = w_480 = 1, else 2; zero for k > 480 or n > 32. */
int seld_gcc_table_host(uint16_t* table);

/* ---- labels: dataset.py:60-119 metadata_to_labels + utils.py:77-90 polar_to_grid ---------- */
/* events: int32 [R][5] = (meta_frame, class, source, azimuth_deg, elevation_deg), the CSV rows after
 * the reference's int() casts (dataset.py:93-97).  T = int((L/sr*1000)/20) label frames, computed by
 * the caller with the reference's float64 rule (dataset.py:73).  mask: uint16 [T][I*J], overwritten:
 * bit c set <=> labels[t, cell, c] == 1.0 from an event row; the background one-hot (class 13 where
 * no event, dataset.py:114-117) is implied by mask == 0.  Rows with 5*meta_frame >= T are dropped
 * (empty range in the reference); rows with class outside [0,16) are ignored (the reference raises
 * IndexError for class >= 14 -- the Python host checks that before upload).  I*J must be even. */
int seld_labels_rasterise(const int32_t* events, int64_t R, int64_t T, int I, int J, uint16_t* mask, void* stream);

/* Gaussian-region label augmentation, smrl_seld_gaussian.py:397-534 (active there, absent from the modular
 * dataset.py).  Like seld_labels_rasterise, but every row paints its class into all cells whose centre lies in the
 * +-2 sigma box around centres[r] = (azimuth + az_noise, elevation + el_noise) (float64, degrees; the per-source
 * noise is drawn on the host, :426-437); azimuth wraps, elevation is clipped to [-90, 90]. */
int seld_labels_rasterise_box(const int32_t* events, const double* centres, int64_t R, int64_t T, int I, int J,
                              double sigma_az, double sigma_el, uint16_t* mask, void* stream);

/* mask uint16 [n_cells] -> dense float32 [n_cells][num_classes] exactly as dataset.py:110-117 leaves it. */
int seld_labels_expand(const uint16_t* mask, int64_t n_cells, int num_classes, float* dense, void* stream);

/* ---- windows: dataset.py:267-317 _create_windows ------------------------------------------ */
/* dst[b][w][:] = src[starts[b] + w][:] (rows of row_bytes, a multiple of 16) for rows inside
 * [0, total_rows), zero bytes otherwise: a zero spectrogram pad (dataset.py:293-294) and mask 0 =
 * background for the labels (dataset.py:298-299).  starts: int64 [B] on the device. */
int seld_window_gather(const void* src, int64_t total_rows, int64_t row_bytes, const int64_t* starts, int64_t B,
                       int64_t window, void* dst, void* stream);

/* ---- loss: loss.py:43-54 class_mse_loss (+ its backward) ---------------------------------- */
/* logits [n_cells][14] (fp32, or bf16 when logits_is_bf16), labels as EITHER the compact mask
 * (uint16 [n_cells]) OR dense float32 [n_cells][14] (exactly one non-NULL).  Writes
 * loss_out[0] = mean((softmax(logits) - y)^2) and, if grad != NULL, grad (same dtype/shape as logits)
 * = grad_scale * p_k * ((p_k - y_k) - sum_c p_c (p_c - y_c)); pass grad_scale = 2*w/(n_cells*14) for
 * d(w*loss)/dlogits.  workspace: seld_softmax_mse_workspace_bytes() bytes of device scratch.
 * Deterministic (fixed-order reduction, no float atomics). */
int64_t seld_softmax_mse_workspace_bytes(void);
int seld_softmax_mse(const void* logits, int logits_is_bf16, const uint16_t* mask, const float* dense_labels,
                     int64_t n_cells, int num_classes, float grad_scale, float* loss_out, void* grad,
                     void* workspace, void* stream);

/* data[0..n) *= *scale with the scale read from DEVICE memory (the upstream gradient autograd hands the loss):
 * when it is exactly 1.0 -- what loss.backward() passes -- every block returns after one load and the data is not
 * touched.  n must be a multiple of 8; data bf16 when is_bf16, else fp32, 16-byte aligned. */
int seld_scale_by_device_scalar(void* data, int is_bf16, int64_t n, const float* scale, void* stream);

/* The optimiser step of trainer.py:112-116,179 (torch.optim.Adam with weight_decay as L2 added to the gradient; amsgrad off)
 * for a LIST of tensors in one launch per 48 tensors, arithmetic of the framework's fused kernel in fp32:
 *   g = grad * grad_scale + weight_decay * p;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;
 *   p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * grad[k]: bf16 when grad_is_bf16[k] (the gradient of a bf16 working weight: no separate cast), else fp32; param / exp_avg /
 * exp_avg_sq fp32; low_bf16[k]: the bf16 working copy rewritten from the new master, or NULL.  All tensors of an entry
 * share one memory layout (the kernel walks the storage).  lr and step are DEVICE scalars (step = the count of THIS
 * update, i.e. already incremented).  HOST arrays of device addresses / lengths, passed to the kernel by value. */
int seld_multi_adam(const void* const* grad, const int32_t* grad_is_bf16, float* const* param, float* const* exp_avg,
                    float* const* exp_avg_sq, void* const* low_bf16, const int64_t* lengths, int count, const float* lr,
                    const float* step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                    void* stream);

/* One launch per 96 tensors casts a list of tensors (the fp32-master / bf16-working-weight mode of the trainer):
 * src / dst are HOST arrays of `count` device addresses, lengths a HOST array of element counts (the descriptors are
 * passed to the kernel by value).  bf16_to_fp32 != 0: bf16 sources -> fp32 destinations (gradients); 0: fp32 -> bf16
 * (weights), round to nearest even.  Source and destination of a pair must share their memory layout (the cast walks
 * the storage).  No reference counterpart: replaces the per-parameter casts of torch.autocast. */
int seld_multi_cast(const void* const* src, void* const* dst, const int64_t* lengths, int count, int bf16_to_fp32,
                    void* stream);

/* LayerNorm over the last dimension [-> ReLU], activations in their own dtype (csrc/layernorm.hip).  Replaces the
 * head's nn.LayerNorm(512) -> nn.ReLU of model_crnn.py:77-83 (model_conformer.py / resnet50_model.py: LayerNorm(1024)).
 * x, y, dy, dx [rows][D] contiguous, fp32 or bf16 (is_bf16); weight, bias [D] fp32; statistics and arithmetic in fp32
 * (biased two-pass variance, eps inside the square root: torch.nn.functional.layer_norm).  D in {256, 512, 1024, 2048}
 * (seld_layernorm_supported).  forward writes mean_rstd [rows][2] fp32 -- with x, all the backward pass needs (the
 * ReLU mask is recomputed).  backward writes dx, dweight [D], dbias [D] (fp32, deterministic two-level sums);
 * workspace: seld_layernorm_workspace_floats(rows, D) floats. */
int seld_layernorm_supported(int64_t D);
int64_t seld_layernorm_workspace_floats(int64_t rows, int64_t D);
int seld_layernorm_forward(const void* x, int is_bf16, int64_t rows, int64_t D, const float* weight, const float* bias,
                           float eps, int relu, void* y, float* mean_rstd, void* stream);
int seld_layernorm_backward(const void* x, const void* dy, int is_bf16, int64_t rows, int64_t D, const float* weight,
                            const float* bias, const float* mean_rstd, int relu, void* dx, float* dweight,
                            float* dbias, float* workspace, void* stream);

/* Hold `stream` for `nanoseconds` (0 .. 1e6) with a one-wavefront kernel that watches the 100 MHz wall clock.
 * Used at the head of the side stream that carries weight-gradient GEMMs beside a BiGRU recurrence
 * (seld_gru_backward): the recurrence's 16 workgroups each need a whole CU's LDS and must be resident before the
 * GEMMs occupy every CU.  No upstream counterpart (the reference's trainer.py:165-179 is single-stream). */
int seld_stream_delay(int64_t nanoseconds, void* stream);

/* The three-term SMR-SELD loss of smrl_seld_gaussian.py:946-1072 (loss.py:43-54, 56-146 on probabilities), value and
 * gradient in one pass (csrc/loss3.hip):  total = w_class * MSE(softmax(logits), y) + w_aiur * AIUR + w_cl * CL.
 * logits [frames][rows*cols][14] fp32 or bf16; labels as seld_softmax_mse (uint16 mask per cell, or dense fp32);
 * rows x cols = the I x J DOA grid (18 x 36), at least 3 x 3 and at most 1024 cells.  loss_out4 (device) receives
 * (total, mse, aiur, cl).  grad (nullable, dtype of logits) = d total / d logits for an upstream gradient of 1 (the AIUR
 * term is argmax based and has none).  workspace: seld_smr_loss_workspace_bytes(frames, rows * cols) bytes.
 * Deterministic (fixed-order double partial sums, integer event counts). */
int64_t seld_smr_loss_workspace_bytes(int64_t frames, int64_t cells_per_frame);
int seld_smr_loss(const void* logits, int logits_is_bf16, const uint16_t* mask, const float* dense_labels,
                  int64_t frames, int rows, int cols, int num_classes, float w_class, float w_aiur, float w_cl,
                  float* loss_out4, void* grad, void* workspace, void* stream);

/* ---- glue kernels of the training iteration (csrc/glue.hip): each replaces a chain of 3-10 framework launches of a
 * few microseconds (clone / fill / add / cast, fill + reduce + copy, slice copies, flip + copy) by one launch ---- */

/* nn.GRU's biases (model_crnn.py:65-72) as the recurrence consumes them: gi_bias [2][3][H] (bf16 when out_is_bf16,
 * else fp32) = b_ih + (r, z rows of b_hh), the bias of the input-projection GEMM; b_hn [2][H] fp32 = n rows of b_hh.
 * b_ih, b_hh: [2][3H] fp32 (forward, reverse). */
int seld_gru_fold_bias(const float* b_ih, const float* b_hh, int64_t H, void* gi_bias, int out_is_bf16, float* b_hn,
                       void* stream);

/* seld_gru_backward's per-tile bias sums `partial` [tiles][2][4][H] -> nn.GRU's bias gradients db_ih [2][3H] =
 * (da_r, da_z, da_n) and db_hh [2][3H] = (da_r, da_z, da_n r), fp32, tiles added in a fixed order. */
int seld_gru_bias_grads(const float* partial, int64_t tiles, int64_t H, float* db_ih, float* db_hh, void* stream);

/* out[i] = sum over c of partial[c][i], i < count (fp32 accumulation, fixed order): the reduction behind a split-K
 * weight-gradient product (the autograd dW = dY^T X of every nn.Linear, trainer.py:178).  partial [chunks][count] bf16
 * or fp32, out [count] bf16 or fp32. */
int seld_sum_chunks(const void* partial, int in_is_bf16, int64_t chunks, int64_t count, void* out, int out_is_bf16,
                    void* stream);

/* out[n] = sum over r of g[r][n]: the bias gradient of an nn.Linear (autograd for trainer.py:178) from the [rows][n_cols]
 * output gradient, bf16 or fp32, n_cols % 8 == 0, 16-byte aligned.  Two launches: row blocks summed in parallel into
 * `partial` [seld_column_sums_blocks(rows, n_cols)][n_cols] fp32 (caller-owned), then added in a fixed order. */
int64_t seld_column_sums_blocks(int64_t rows, int64_t n_cols);
int seld_column_sums(const void* g, int in_is_bf16, int64_t rows, int64_t n_cols, float* partial, void* out, int out_is_bf16,
                     void* stream);

/* Multi-tensor forms of the two reductions above, for a backward pass that queues them (every nn.Linear of
 * model_conformer.py:19-41,98-127 / resnet50_model.py:80-91 contributes one of each; nothing reads a weight or bias
 * gradient before the optimiser of trainer.py:179): ONE launch per 64 chunk sums, TWO (row-block partials, fixed-order
 * finish) per 40 column sums, descriptors by value.  flags[k]: bit 0 = the input is bf16, bit 1 = the output is bf16.
 * seld_multi_column_sums needs caller-owned fp32 scratch `partial` sized by seld_multi_column_sums_scratch. */
int seld_multi_sum_chunks(const void* const* partial, void* const* out, const int64_t* counts, const int32_t* chunks,
                          const int32_t* flags, int count, void* stream);
int seld_multi_column_sums_scratch(const int64_t* rows, const int64_t* n_cols, int count, int64_t* partial_floats);
int seld_multi_column_sums(const void* const* g, void* const* out, const int64_t* rows, const int64_t* n_cols,
                           const int32_t* flags, int count, float* partial, int64_t partial_floats, void* stream);

/* dW_hh [2][3H][H] of nn.GRU from the two chunked products the host forms over both directions at once:
 * p_gi [chunks][2][3][H][2][H] = (da_r, da_z, da_n)^T h_prev, p_n [chunks][2][H][2][H] = (da_n r)^T h_prev; the blocks
 * with matching directions (and, of p_gi, the r and z gates) are summed over the chunks and written in place. */
int seld_gru_dwhh_finish(const void* p_gi, const void* p_n, int in_is_bf16, int64_t chunks, int64_t H, void* dw_hh,
                         int out_is_bf16, void* stream);

/* wt[i][o][2-r][2-s] = w[o][i][r][s] for 3x3 weights, both in channels-last memory (w: [O][3][3][I], wt: [I][3][3][O]):
 * the weights with which the DATA gradient of a 3x3 / stride 1 / pad 1 convolution (model_crnn.py:5-17) is itself a
 * forward convolution.  elem_bytes 2 (bf16) or 4. */
int seld_conv_weight_flip_transpose(const void* w, int elem_bytes, int64_t O, int64_t I, void* wt, void* stream);

/* ---- CNN block tail: BatchNorm2d -> ReLU -> MaxPool2d((1,2)) at model_crnn.py:5-17 (ConvBlock.forward) ---- */
/* x: the convolution output in channels-last memory order = row-major [rows = B*T*F][C] (bf16 when is_bf16, else
 * fp32); the two frequency bins of a pooling pair are adjacent rows.  pool = 2: MaxPool2d((1,2)); pool = 1: no
 * pooling; pool = 4: no pooling and SiLU instead of ReLU (BatchNorm1d -> Swish of the Conformer convolution module,
 * model_conformer.py:71-96, rows = B*T).  C must be 8 * (a divisor of 256); rows even when pool = 2.
 * forward, training != 0: batch statistics (biased variance), running_mean / running_var updated with `momentum`
 *   (unbiased variance) exactly like nn.BatchNorm2d; training == 0: the running statistics are used.
 *   y [rows/pool][C] (dtype of x) = max over the pair of relu(weight * (x - mean) * invstd + bias), with the
 *   BatchNorm output rounded to the activation dtype before the comparisons like the unfused modules.
 *   mean_invstd [2][C], scale_shift [2][C] (a = weight*invstd, b = bias - mean*a): outputs, kept for backward.
 * backward: dx [rows][C] (dtype of x), dweight [C], dbias [C] from dy [rows/pool][C]; the ReLU mask and pooling
 *   argmax are recomputed from x (first element wins ties, as max_pool2d).
 * workspace: seld_conv_tail_workspace_floats(C) floats of device scratch.  Deterministic (no float atomics). */
int64_t seld_conv_tail_workspace_floats(int C);
int seld_conv_tail_forward(const void* x, const void* residual, int is_bf16, int64_t rows, int C, int pool,
                           const float* weight, const float* bias, float* running_mean, float* running_var,
                           float momentum, float eps, int training, void* y, float* mean_invstd, float* scale_shift,
                           float* workspace, void* stream);
int seld_conv_tail_backward(const void* x, const void* residual, const void* dy, int is_bf16, int64_t rows, int C,
                            int pool, const float* mean_invstd, const float* scale_shift, void* dx, void* dresidual,
                            float* dweight, float* dbias, float* workspace, void* stream);

/* ---- Conformer convolution module: depthwise Conv1d over time, model_conformer.py:71-96 ------------------- */
/* nn.Conv1d(D, D, K, padding=(K-1)/2, groups=D) evaluated in the channels-last layout of the surrounding layers:
 * x, y [B][T][D] (bf16 when is_bf16, else fp32), weight [D][K] fp32, bias [D] fp32 or NULL, odd K <= 31, D % 64 == 0.
 * flip_taps != 0 evaluates the data gradient (dx from dy with the taps reversed; pass bias = NULL).
 * seld_dwconv1d_wgrad: partial [R][D][32] fp32 with R = seld_dwconv1d_wgrad_rows(B, T) (one row per batch row and
 * 50-step time chunk) -- slots 0..K-1 = sum_t dy[t] x[t + k - pad] over the chunk, slot 31 = sum_t dy[t]; the caller adds
 * the rows (dweight = partial.sum(0)[:, :K], dbias = partial.sum(0)[:, 31]). */
int seld_dwconv1d(const void* x, int is_bf16, const float* weight, const float* bias, int64_t B, int64_t T, int D, int K,
                  int flip_taps, void* y, void* stream);
int64_t seld_dwconv1d_wgrad_rows(int64_t B, int64_t T);
int seld_dwconv1d_wgrad(const void* x, const void* dy, int is_bf16, int64_t B, int64_t T, int D, int K, float* partial,
                        void* stream);

/* ---- recurrence: nn.GRU(2048, 256, num_layers=2, bidirectional) at model_crnn.py:65-72 ------- */
/* One bidirectional GRU layer's recurrence, all T steps in one launch (both directions), H = 256.
 * The batch is processed in tiles of S = seld_gru_tile_rows() sequences (4 in this build; tiles = ceil(B / S)):
 * one workgroup per (tile, direction); the other columns of each 16-column MFMA are padding so that the
 * per-CU vector-memory and gate-math work of a step -- what bounds it once the weights are resident -- is
 * spread over more CUs.
 *
 * Streamed per-step tensors use the kernels' private TILE LAYOUT so that every wavefront load / store
 * is one contiguous run: with P = 16/S lanes sharing a sequence and U = 8/P units per lane, a tensor
 * X[b][t][dir][slot][u] (b = S*tile + seq, u = 32*w + 16*s + 4*q + i, 4*s + i = U*part + j) is stored as
 * [tile][t][dir][w(8)][slot(NS)][q(4)][part(P)][seq(S)][j(U)]  (seld_gru_to_tile / seld_gru_from_pair_tile
 * convert; seld_native.to_tile / from_tile are their torch definitions).
 *   gi         [B][T][2][3][H] NATURAL layout -- the input GEMM's output as it is (no permute: the loads are 4 bytes
 *                    per lane either way): x W_ih^T + b_ih, gates r|z|n, direction 0 = forward in time,
 *                    1 = reverse (fp32, or bf16 when is_bf16).  The recurrent biases of the r and z gates
 *                    (b_hh[0:2H]) must ALREADY be added in (they commute with the sigmoid argument).
 *   w_hh       [2][3H][H] bf16;   b_hn [2][H] fp32 (= b_hh[2H:3H] per direction);   h0 = 0
 *   y          [tiles*S][T][2H] natural layout (h_t; forward direction in [..., :H]), dtype of gi; rows >= B are
 *                    scratch of the last tile's padding sequences (which re-read sequence B-1's gi)
 *   saved_tile       r, z, n, (W_hn h + b_hn) per step for the backward pass, or NULL: 2 pair-slots (r|z, n|gh_n),
 *                    i.e. [tile][t][dir][w(8)][2][lane(64)][2][U]; fp32 when gi is fp32, IEEE fp16 when is_bf16
 *                    (O(1) values: 8x finer than bf16 at half of fp32's bytes -- the recurrence is bound by one
 *                    CU's load/store path).  Opaque to the caller: tiles*T*2*8*2*64*2*U elements.
 * MFMA bf16 operands, fp32 accumulation, fp32 gates and state. */
int64_t seld_gru_tile_rows(void);
int seld_gru_forward(const void* gi, int is_bf16, const void* w_hh_bf16, const float* b_hn, int64_t B,
                     int64_t T, int64_t H, void* y, void* saved_tile, void* stream);

/* Backward of the recurrence.  dy_tile NS=1 (dtype of y), y = the forward output (all tiles*S rows),
 * w_hh_t [2][H][3H] bf16 (W_hh transposed).
 * dg_tile (dtype of y): da_r, da_z, da_n, da_n*r as 2 pair-slots (da_r|da_z, da_n|da_n*r), i.e.
 * [tile][t][dir][w(8)][2][q(4)][part(P)][seq(S)][2][j(U)] -- the first three are d/d(gi); (da_r, da_z, da_n*r) are
 * d/d(gh), from which the caller forms dW_hh = dgh^T h_prev and with gi's GEMM dW_ih, dx.
 * dbias [tiles][2][4][H] fp32: the four slots summed over the tile's sequences and all t (add the tiles for
 * db_ih = slots (0,1,2) and db_hh = slots (0,1,3)). */
int seld_gru_backward(const void* dy_tile, const void* saved_tile, const void* y, int is_bf16,
                      const void* w_hh_t_bf16, int64_t tiles, int64_t T, int64_t H, void* dg_tile, float* dbias,
                      void* stream);

/* Layout converters for the two calls above (HBM-bound permutes; elem_bytes = 2 for bf16, 4 for fp32).
 * seld_gru_to_tile: natural src [B][T][2][ns][H] -> tile layout dst (B padded with zeros to whole tiles).
 * seld_gru_from_pair_tile: the backward kernel's dg_tile -> dgi [B][T][2][3][H] (da_r, da_z, da_n) and
 * dghn [B][T][2][H] (da_n*r), both contiguous -- what the caller's GEMMs read. */
int seld_gru_to_tile(const void* src, int elem_bytes, int64_t B, int64_t T, int ns, void* dst, void* stream);
int seld_gru_from_pair_tile(const void* dg_tile, int elem_bytes, int64_t B, int64_t T, void* dgi, void* dghn,
                            void* stream);

/* h_{t-1} of the forward recurrence, natural layout: y [B][T][2][H] (the forward output, h_t) ->
 * h_prev[b][t][0] = y[b][t-1][0], h_prev[b][t][1] = y[b][t+1][1], zero at each direction's first step.  The operand of
 * the recurrent weight gradient dW_hh = sum_t dgh_t^T h_{t-1} (what autograd through model_crnn.py:65-72's nn.GRU
 * accumulates step by step). */
int seld_gru_previous_state(const void* y, int elem_bytes, int64_t B, int64_t T, void* h_prev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SELD_HIP_H_ */
