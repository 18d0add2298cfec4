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
int seld_init(int device);            /* builds the constant tables (Hann, twiddles, sparse mel) */
int seld_shutdown(void);
int seld_version(void);
const char* seld_last_error(void);

/* Override the default HTK mel filterbank with a host table fb[481][64] (fp32), e.g. the one
 * torchaudio.functional.melscale_fbanks would build.  Replaces the constant inside
 * torchaudio.transforms.MelScale used at dataset.py:38-43. */
int seld_set_mel_filterbank(const float* fb_host);

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

#ifdef __cplusplus
}
#endif
#endif /* SELD_HIP_H_ */
