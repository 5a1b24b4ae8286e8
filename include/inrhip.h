/*
 * inrhip.h -- C ABI of libinrhip.so: the MI355X (gfx950) kernels for the INR / SIREN
 * super-resolution fit path of MRIRC/MRI-super-resolution.
 *
 * The reference has no FFI of its own: its boundary is the Python module surface of SRDWI.py /
 * INRmodel.py / nn_mri.py, and all arithmetic is stock PyTorch ops.  Every entry point below
 * therefore replaces a *framework op sequence* of the reference; the file:line of that sequence
 * (relative to /root/reference/implicit-neural-representations) is cited per function.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, int64 sizes, scalar hyper-parameters, a hipStream_t passed as
 *    void* (0 = the null stream).  No torch types, no ownership transfer: the caller allocates every
 *    output and every workspace (size from the matching *_workspace_bytes query).
 *  - every function only ENQUEUES work on `stream` and returns without synchronising.  Everything a call launches goes onto
 *    `stream` itself, in program order: the library owns NO stream and creates no events of its own for ordering (rounds 1-3
 *    forked the parameter-gradient GEMMs onto a library-owned side stream; since round 4 they are one merged launch on
 *    `stream`).  Consequently (a) whatever the caller enqueues on `stream` after the call is ordered behind all of its work,
 *    (b) calls on DIFFERENT streams are not ordered against each other by the library, and (c) the calls contain no
 *    allocation and no host synchronisation, so a sequence of them can be captured into a HIP graph.  TWO documented
 *    exceptions to "no host sync": when inr_siren_fit / inr_siren_fit_cycle take the persistent cooperative small-network
 *    kernel (hidden 32 / 64, <= 32 input features, one output, few thousand rows) they wait for `stream` once at the end of
 *    the call to read the kernel's completion word, and return INR_E_TIMEOUT if a launch was abandoned (cooperative
 *    launches cannot be captured into a graph in any case); and inr_prof_read() waits for the events it reports on.
 *  - threading / concurrency: entry points may be called concurrently from several host threads on different streams with
 *    DISJOINT output and workspace buffers (read-only inputs may be shared); this is what
 *    drivers.run_volumes(concurrent=k) relies on -- k fits of one process, each on a host thread and a stream of its own,
 *    each with its own parameter / moment / workspace buffers.  One call uses one workspace; the same workspace must not be
 *    in use by two calls that may overlap on the device.  The cooperative small-network kernel needs its whole grid
 *    co-resident: two such launches on different streams at the same time can each hold part of the chip and stall until the
 *    grid barrier's poll limit reports INR_E_TIMEOUT -- run those fits one after the other (run_volumes does).
 *    Process-global state, all of it safe to touch from several threads, none of it carrying tensor data:
 *      (i)   the thread-local message behind inr_last_error();
 *      (ii)  the event profiler behind inr_prof_* (a mutex; off unless enabled);
 *      (iii) the launch counters behind inr_launch_count (atomics);
 *      (iv)  the workspace stamps that guard INR_REUSE_* flags and the pending inr_siren_forward_train stash: 64 entries
 *            keyed on the workspace ADDRESS (a mutex; least recently used entry replaced).  A stamp says which (n, x,
 *            target, weight) the operand image in that workspace was built from; it cannot know that the caller freed the
 *            workspace and received the same address again -- a caller that recycles workspace memory must not pass
 *            INR_REUSE_* on the first call after doing so;
 *      (v)   the per-device co-residency limits of the cooperative kernel (atomics, computed once per device);
 *      (vi)  the DIAGNOSTIC switches behind inr_debug_set / inr_debug_set_ptr, which select kernel families for A/B
 *            measurements and tests.  They are atomics, but process-wide: flipping one while another thread is enqueueing
 *            changes what that thread launches next (and whether inr_siren_hp_eligible says yes).  A production caller never
 *            touches them; a test harness restores them with inr_debug_reset().
 *  - return value: 0 = ok; negative = invalid argument (INR_E_*); positive = hipError_t.
 *    inr_last_error() returns a thread-local human-readable message for the last failure.
 *  - all tensors are dense row-major fp32.  Linear weights are [out_features][in_features] exactly as
 *    torch.nn.Linear stores them.
 *  - reductions (loss, bias/weight gradients) use fixed-order two-stage sums: no float atomics, so
 *    two runs on the same inputs are bitwise identical.
 */
#ifndef INRHIP_H
#define INRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INR_ABI_VERSION 1

#define INR_E_INVALID   (-1)  /* null pointer / non-positive size / unsupported shape            */
#define INR_E_WORKSPACE (-2)  /* workspace pointer null or smaller than the *_workspace_bytes query */
#define INR_E_ALIGN     (-3)  /* pointer not 16-byte aligned where the kernel requires it         */
#define INR_E_TIMEOUT   (-4)  /* a persistent kernel abandoned its launch (grid barrier poll limit): outputs of the call are
                                 partly written and must be discarded                                */

typedef struct inr_device_caps {
    int  abi_version;
    int  device;
    int  compute_units;
    int  wavefront_size;
    int  lds_bytes_per_cu;
    int  clock_khz;
    int64_t hbm_bytes;
    char arch[32];          /* "gfx950..." */
} inr_device_caps_t;

/* Description of one SIREN: Siren(in, hidden, hidden_layers, out, first_omega_0, hidden_omega_0)
 * (SRDWI.py:67-85).  Sine layers: 1 + hidden_layers; then the linear head. */
typedef struct inr_siren_desc {
    int   in_features;
    int   hidden_features;
    int   hidden_layers;
    int   out_features;
    float first_omega;
    float hidden_omega;
} inr_siren_desc_t;

/* RAMS multi-image network (multi-image-super-resolution/utils/network.py:91-155): RAMS(scale, filters, kernel_size,
 * channels, r, N).  The kernels are written for filters == 32 and kernel_size == 3 (the reference's only
 * configuration, multi-image-super-resolution/master.py:20-25). */
typedef struct inr_rams_desc {
    int   scale;        /* 3 */
    int   filters;      /* 32 */
    int   kernel_size;  /* 3 */
    int   channels;     /* T = 9 acquisitions per stack */
    int   r;            /* squeeze ratio 8 */
    int   n_rfab;       /* N = 12 */
    float mean;         /* 7433.6436 (network.py:18) */
    float std;          /* 2353.0723 (network.py:19) */
} inr_rams_desc_t;

/* Flat parameter layout used by the fused entry points (network order):
 *   W_0[hidden][in], b_0[hidden], W_1[hidden][hidden], b_1, ..., W_head[out][hidden], b_head[out]
 * Every tensor starts at a multiple of 4 floats (16 B): offsets come from inr_siren_param_offsets. */

int         inr_version(void);
/* 0 for the product build.  Non-zero = a diagnostic build (bit 0: in-kernel time stamps -DINR_STAMPS, bit 1: ablated
 * kernels -DH3_ABLATE, bit 2: padded LDS -DH3_EXTRA_LDS): its timings and, with bit 1, its RESULTS are not the product's.
 * The Python binding refuses such a library unless it was selected explicitly (INR_LIB=...). */
int         inr_build_flags(void);
const char* inr_last_error(void);
int         inr_device_caps(int device, inr_device_caps_t* out);

/* ---- a-1: get_mgrid (SRDWI.py:12-18, nn_mri.py:87-94) ------------------------------------------
 * rows [row_begin, row_begin+n_rows) of the flattened 'ij' meshgrid of linspace(-1,1,shape[a]),
 * last axis fastest; bit-exact with torch.linspace (single-rounding fma rule).  dim<=8. */
int inr_mgrid(float* out, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows, void* stream);

/* ---- a-3: input_mapping (SRDWI.py:111-116) -----------------------------------------------------
 * out[n][2m] = [sin(2*pi*x @ B^T) | cos(2*pi*x @ B^T)],  x[n][d], B[m][d]. */
int inr_fourier_map(float* out, const float* x, const float* B, int64_t n, int d, int m, void* stream);
/* Same, with x generated in-kernel from the grid (K1+K2 fused: the coordinate grid never reaches HBM;
 * replaces get_mgrid(...).cuda() -> input_mapping at superresDWI.py:125-126). */
int inr_grid_fourier_map(float* out, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows,
                         const float* B, int m, void* stream);

/* ---- a-4: SineLayer.forward (SRDWI.py:58-59): act = sin(omega*(x W^T + b)) ----------------------
 * x[n][in], W[out][in], b[out] (nullable), act[n][out]; dact (nullable) receives
 * omega*cos(omega*(x W^T + b)) -- the factor autograd multiplies by in backward. */
int inr_sine_layer_forward(float* act, float* dact, const float* x, const float* W, const float* b,
                           int64_t n, int in_features, int out_features, float omega, void* stream);

/* ---- a-10: PerturbNet layers (SRDWI.py:93-109) ---------------------------------------------------------
 * hidden layer:  act = scale*tanh(x W^T + b), dact (nullable) = scale*(1 - tanh^2)   [same GEMM as the sine layer]
 * output layer:  y[n][out] = scale*tanh(a W^T + b) (scale = eps, SRDWI.py:107), dy (nullable) its derivative;
 *                row-dot per output with a wavefront shuffle reduction (out = d = 2..4 columns).
 * The constant acquisition column of SRDWI.py:102-104 is folded into the bias by the caller:
 * b_eff = b + (sample/10) * W[:, in]  (so x stays the [n, in] feature matrix and nothing is concatenated). */
int inr_tanh_layer_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n,
                           int in_features, int out_features, float scale, void* stream);
int inr_linear_tanh_head_forward(float* y, float* dy, const float* a, const float* W, const float* b, int64_t n,
                                 int in_features, int out_features, float scale, void* stream);

/* element-wise out = a*b over `count` floats: dz = grad_out * dact, the first step of a stand-alone
 * SineLayer's backward (what autograd does for torch.sin(omega*z), SRDWI.py:59).  out may alias a or b. */
int inr_mul(float* out, const float* a, const float* b, int64_t count, void* stream);

/* ---- a-5: final nn.Linear (SRDWI.py:75-77,83): y = a W^T + b, optional clamp(min) ---------------
 * (clamp fuses torch.clamp(..., min=0) of superresDWI.py:161; pass use_clamp=0 for the raw head). */
int inr_linear_head_forward(float* y, const float* a, const float* W, const float* b, int64_t n,
                            int in_features, int out_features, int use_clamp, float clamp_min, void* stream);

/* ---- a-6: ((y-t)**2).mean() and its gradient (superresDWI.py:135; weighted: master.py:143-145) ---
 * gy[i] = 2*w[i]*(y[i]-t[i])/count ; *loss = mean(w*(y-t)^2).  w nullable.  count = n*out. */
size_t inr_mse_workspace_bytes(int64_t count);
int inr_mse_loss_grad(float* gy, float* loss, const float* y, const float* t, const float* w,
                      int64_t count, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a-6: backward pieces (what autograd runs for SRDWI.py:58-59,83) -----------------------------
 * head:   dz_last[n][hidden] = (gy[n][out] @ W_head[out][hidden]) * dact_last   (in place over dact ok)
 *         gW_head[out][hidden] = gy^T a_last ; gb_head[out] = colsum(gy)
 *         gb_last[hidden] (nullable, needs dz_last) = colsum(dz_last): the bias gradient of the last sine
 *         layer, produced by the same pass (one fused kernel when out_features == 1). */
size_t inr_head_backward_workspace_bytes(int64_t n, int hidden, int out_features);
int inr_linear_head_backward(float* dz_last, float* gW, float* gb, float* gb_last, const float* gy,
                             const float* a_last, const float* dact_last, const float* W, int64_t n, int hidden,
                             int out_features, void* workspace, size_t workspace_bytes, void* stream);
/* input grad through one sine layer: dz_prev[n][in] = (dz[n][out] @ W[out][in]) * dact_prev[n][in]
 * (dz_prev may alias dact_prev).  With dact_prev == NULL writes the plain product dz @ W.
 * gb_prev[in] (nullable) = colsum(dz_prev), the bias gradient of the layer below, accumulated in the GEMM
 * epilogue; it needs the workspace (otherwise workspace may be NULL). */
size_t inr_sine_layer_backward_input_workspace_bytes(int64_t n, int in_features);
int inr_sine_layer_backward_input(float* dz_prev, float* gb_prev, const float* dz, const float* W,
                                  const float* dact_prev, int64_t n, int in_features, int out_features,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* parameter grads: gW[out][in] = dz^T x ; gb[out] = colsum(dz).  Split over rows + fixed-order reduce. */
size_t inr_linear_param_grad_workspace_bytes(int64_t n, int in_features, int out_features);
int inr_linear_param_grad(float* gW, float* gb, const float* dz, const float* x, int64_t n,
                          int in_features, int out_features, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ---- a-7: torch.optim.Adam.step (superresDWI.py:116,138; defaults b=(0.9,0.999), eps=1e-8) ------
 * One launch over `count` contiguous fp32 elements.  bias corrections are computed on the host in
 * double exactly like torch's single-tensor path: step_size = lr/(1-b1^t), denom =
 * sqrt(v)/sqrt(1-b2^t) + eps.  `step` is the 1-based step number. */
int inr_adam_step(float* p, const float* g, float* m, float* v, int64_t count, int64_t step,
                  double lr, double beta1, double beta2, double eps, void* stream);

/* ---- fused SIREN entry points (flat parameter buffer) ------------------------------------------ */
int64_t inr_siren_param_count(const inr_siren_desc_t* desc);      /* padded flat length in floats */
/* offsets[2*(hidden_layers+2)]: (W_l, b_l) float offsets in network order, head last */
int  inr_siren_param_offsets(const inr_siren_desc_t* desc, int64_t* offsets);

/* a-5 + a-9: y[n][out] = head(sine layers(x[n][in])), optional clamp; forward only, no stash. */
size_t inr_siren_forward_workspace_bytes(const inr_siren_desc_t* desc, int64_t n);
int inr_siren_forward(const inr_siren_desc_t* desc, const float* params, const float* x, int64_t n,
                      float* y, int use_clamp, float clamp_min, void* workspace, size_t workspace_bytes,
                      void* stream);

/* a-9: dense re-sampling clamp(INR(input_mapping(get_mgrid(shape), B)), min) -> y[prod(shape)][out]
 * (superresDWI.py:125-126,161-162; superresHybrid.py:103-104,119).  Grid + Fourier features are
 * generated chunk by chunk in the workspace (never the whole [N_test,2m] feature matrix).
 * B == NULL means raw coordinates feed the network (nn_mri path, master.py:149-153). */
size_t inr_siren_reconstruct_workspace_bytes(const inr_siren_desc_t* desc, int64_t chunk_rows);
int inr_siren_reconstruct(const inr_siren_desc_t* desc, const float* params, const int64_t* shape, int dim,
                          const float* B, int m, float* y, int use_clamp, float clamp_min,
                          int64_t chunk_rows, void* workspace, size_t workspace_bytes, void* stream);

/* a-8: `n_steps` full-batch fit steps (superresDWI.py:132-138; superresHybrid.py:109-114):
 * forward with stash -> MSE (+optional weights) -> backward -> Adam, all enqueued on `stream`,
 * no host sync.  params/grads/m/v are flat buffers of inr_siren_param_count floats.  losses[n_steps]
 * (device, nullable) receives the loss of every step.  first_step is the 1-based Adam step of the
 * first iteration (so a fit can be continued).
 * Arithmetic: when every sine layer is a multiple of 32 wide, the three dense contractions of a step run on the fp16
 * matrix cores with hi/lo-split operands (three fp16 products per fp32 product, fp32 accumulation, power-of-two
 * tensor scales from exact on-device maxima) -- fp32-class accuracy, same parity tolerances; otherwise
 * (and in every stand-alone layer entry point above) on the f32-input MFMA.  The workspace holds the fp16 weight
 * planes of that path; size it with the query function. */
size_t inr_siren_fit_workspace_bytes(const inr_siren_desc_t* desc, int64_t n);
int inr_siren_fit(const inr_siren_desc_t* desc, float* params, float* grads, float* m, float* v,
                  const float* x, const float* target, const float* weight, int64_t n,
                  int64_t first_step, int n_steps, double lr, double beta1, double beta2, double eps,
                  float* losses, void* workspace, size_t workspace_bytes, void* stream);

/* a-11 / master.py:137-148: the same loop when the target (and weight) image changes every step -- `targets` and
 * `weights` (nullable) hold n_acq images of n*out_features floats back to back, step `it` fits image
 * (first_acq + it) % n_acq.  inr_siren_fit is the n_acq = 1 case.  For small networks (hidden 32 or 64, <= 32 inputs,
 * one output, n <= 16,384) all steps run inside ONE persistent cooperative launch per 64 steps (grid barriers between
 * the backward pass, the fixed-order gradient reduction + Adam, and the next forward): no launch per step at all. */
int inr_siren_fit_cycle(const inr_siren_desc_t* desc, float* params, float* grads, float* m, float* v,
                        const float* x, const float* targets, const float* weights, int n_acq, int first_acq,
                        int64_t n, int64_t first_step, int n_steps, double lr, double beta1, double beta2, double eps,
                        float* losses, void* workspace, size_t workspace_bytes, void* stream);

/* (e) one fit split over several GPUs: forward + loss + backward of THIS rank's row shard, no optimizer.
 * The mean of the loss runs over count_total elements (0 = n*out_features, i.e. an unsplit fit), so gradients and
 * losses of the shards add up to the full-batch step: all-reduce(sum) `grads` (flat, inr_siren_param_count floats)
 * and `loss`, then call inr_adam_step on every rank.  Workspace: inr_siren_fit_workspace_bytes(desc, n). */
#define INR_REUSE_INPUT_IMAGE  1   /* x, n and `workspace` are those of the previous call on this workspace: keep the
                                      operand image of x and its scale (skips two passes over x per call) */
#define INR_REUSE_TARGET_STATS 2   /* target / weight are those of the previous call: keep their maxima */
int inr_siren_loss_grad(const inr_siren_desc_t* desc, const float* params, float* grads, const float* x,
                        const float* target, const float* weight, int64_t n, int64_t count_total, float* loss,
                        void* workspace, size_t workspace_bytes, void* stream);
/* the same with `flags` (INR_REUSE_*): a fit whose rows are split over GPUs calls this once per step on unchanged inputs --
 * without the flags every call re-measures and re-converts x (0.25 ms at 524,288 rows x 256 features).  The caller vouches for
 * "unchanged": same x contents, same n, same workspace, nothing else written to it in between. */
int inr_siren_loss_grad_ex(const inr_siren_desc_t* desc, const float* params, float* grads, const float* x,
                        const float* target, const float* weight, int64_t n, int64_t count_total, float* loss,
                        void* workspace, size_t workspace_bytes, int flags, void* stream);

/* (f) the reference's OWN loop, unmodified (superresDWI.py:132-138): `out = INR(x)` -> torch forms the loss -> `loss.backward()`
 * -> `torch.optim.Adam.step()`.  The autograd Function behind `Siren.forward` (replaces SRDWI.py:87-91 + autograd's backward of
 * it) calls these two instead of the layer-by-layer entry points when inr_siren_hp_eligible(desc) != 0:
 *   inr_siren_forward_train  y[n] = network(x) on the pre-split kernels of the fused fit, every layer's stash kept in `workspace`
 *                            (the last sine layer stashes z + b only); flags: INR_REUSE_INPUT_IMAGE as above.
 *   inr_siren_backward_train grads (flat, network order, inr_siren_param_count floats; every element written) = d(sum_r gy[r] y[r])
 *                            / d params from the stash of the LAST inr_siren_forward_train on this workspace (INR_E_INVALID when
 *                            there is none, or n differs); gy = dL/dy [n] as autograd hands it over.  `params` must still hold
 *                            the values the forward ran on.  One backward per forward.
 * Workspace: inr_siren_fit_workspace_bytes(desc, n).  Both only enqueue.  (What a workspace holds -- a pending forward, an operand
 * image for the INR_REUSE_* flags -- is remembered on the host for the 64 most recently used workspaces of the process: the least recently used entry makes room.) */
int inr_siren_hp_eligible(const inr_siren_desc_t* desc);
int inr_siren_forward_train(const inr_siren_desc_t* desc, const float* params, const float* x, float* y, int64_t n,
                            void* workspace, size_t workspace_bytes, int flags, void* stream);
int inr_siren_backward_train(const inr_siren_desc_t* desc, const float* params, float* grads, const float* gy, int64_t n,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- a-12: end-of-fit metrics on the device (fp64 accumulation, fixed-order reductions) -----------------
 * workspace for all three image metrics: inr_metric_workspace_bytes(n_images).
 * inr_psnr:   out[b] (double) = 10*log10(data_range^2 / mean((x_b - y_b)^2)); x, y are [n_images][per_image] fp32.
 *             (the definition of skimage.metrics.peak_signal_noise_ratio, imported at master.py:14)
 * inr_ssim2d: out[b] (double) = skimage-0.20 structural_similarity(x_b, y_b, data_range) with default arguments
 *             (uniform win x win window, sample covariance, K1=.01, K2=.03, border crop) on [n_images][H][W] fp32;
 *             use_mask != 0 multiplies both images by (x > mask_thr) first -- the protocol of
 *             superresDWI.py:183-186.
 * inr_adc_map: calculate_ADC (SRDWI.py:118-130): out[p] = clip(-slope of lstsq(log(data[p][:] + 1e-7) ~ b/1000), -10, 3)
 *             for data [n_pixels][n_b] fp32, bvals [n_b] fp32 (n_b <= 32). */
size_t inr_metric_workspace_bytes(int n_images);
int inr_psnr(double* out, const float* x, const float* y, int n_images, int64_t per_image, double data_range,
             void* workspace, size_t workspace_bytes, void* stream);
int inr_ssim2d(double* out, const float* x, const float* y, int n_images, int height, int width, int win,
               double data_range, int use_mask, float mask_thr, void* workspace, size_t workspace_bytes,
               void* stream);
int inr_adc_map(float* out, const float* data, const float* bvals, int64_t n_pixels, int n_b, void* stream);

/* ---- a-15 / (f)-4: RAMS training step (utils/training.py:193-209 train_step; utils/loss.py:26-75 l1_loss; weight
 * normalisation utils/network.py:29-35).  Parameters in the reference's own variables ("raw" layout): per layer, in graph
 * order, v [taps * cin][cout] (the TensorFlow kernel [k..., cin, cout] flattened), g [cout], b [cout], every segment padded to
 * 4 floats -- inr_rams_train_param_offsets returns the three offsets per layer (and the layer count).
 * inr_rams_train_grads: forward (every intermediate kept) -> per-image cL1 loss[b] (double) -> d(sum_b loss[b]) / d params
 * into `grads` (same layout); `pred` (nullable) receives the un-clipped prediction [B][scale*H][scale*W].
 * x [B][H][W][9] fp32, y_true / mask [B][scale*H][scale*W] fp32, H == W.
 * inr_rams_train_step: the same, then Adam in Keras' form (p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps)). */
int64_t inr_rams_train_param_count(const inr_rams_desc_t* desc);
int     inr_rams_train_param_offsets(const inr_rams_desc_t* desc, int64_t* offsets, int max_layers);
size_t  inr_rams_train_workspace_bytes(const inr_rams_desc_t* desc, int batch, int height, int width);
int inr_rams_train_grads(const inr_rams_desc_t* desc, const float* params, float* grads, const float* x, const float* y_true,
                         const float* mask, double* loss, float* pred, int batch, int height, int width, void* workspace,
                         size_t workspace_bytes, void* stream);
int inr_rams_train_step(const inr_rams_desc_t* desc, float* params, float* grads, float* m, float* v, const float* x,
                        const float* y_true, const float* mask, double* loss, int batch, int height, int width, int64_t step,
                        double lr, double beta1, double beta2, double eps, void* workspace, size_t workspace_bytes, void* stream);

/* ---- (f)-2: per-voxel acquisition combinations (SRDWI.py:143-152 calculate_combinations, mapped over all voxels by a
 * 32-process pool at superresDWI.py:57-76).  raw_b0 [n_voxels], raw_bk [n_voxels][nk] (k = 1..3) fp32 ->
 * out [n_voxels][4][K], K = n1*n2*n3, combination index in itertools.product order (the last b-value's index fastest). */
int inr_acquisition_products(float* out, const float* raw_b0, const float* raw_b1, const float* raw_b2, const float* raw_b3,
                             int64_t n_voxels, int n1, int n2, int n3, void* stream);

/* ---- (f)-3: the spline baseline every SSIM row is reported against (superresDWI.py:172-191) -----------------------
 * skimage.transform.rescale(img, s, anti_aliasing=True) with its default order (1) for s >= 1: skimage 0.20 hands this
 * to scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True); the anti-aliasing filter has sigma 0 when up-scaling.
 * in [n_images][H][W] fp32 -> out [n_images][OH][OW] fp32; coordinates and weights in double. */
int inr_rescale2d_linear(float* out, const float* in, int n_images, int height, int width, int out_height, int out_width,
                         void* stream);

/* ---- a-13/a-14: RAMS forward + predict_tensor (network.py:91-155, prediction.py:76-83) --------------------------
 * x [B][H][W][channels] fp32 (uint16-range values) -> out [B][scale*H][scale*W] fp32.  clip_round != 0 applies
 * predict_tensor's clip to [0, 2^16] and round-half-to-even.
 * `params` is ONE packed fp32 buffer with weight normalisation already folded (kernel = g*v/||v||, norm over every
 * axis but the output channel); every segment starts at a multiple of 4 floats, in this order:
 *   stem: w[27][32], b[32];
 *   n_rfab x RFAB: conv1 w[27][32][32], b[32]; conv2 w, b; squeeze w[32][32/r], b[32/r]; excite w[32/r][32], b[32];
 *   trunk conv w, b;
 *   (channels/3) x { RFAB as above; reduction conv w[27][32][32], b[32] };
 *   up conv w[27][32][32] (output channels scale^2, zero-padded to 32), b[32];
 *   RTAB: conv1 w[9][T][T], b[T]; conv2 w, b; squeeze w[T][max(T/r,1)], b; excite w[max(T/r,1)][T], b[T];
 *   global conv w[9][T][scale^2], b[scale^2].
 * (conv kernels are in TensorFlow order: tap-major [k1][k2][k3], then input channel, then output channel.) */
int64_t inr_rams_param_count(const inr_rams_desc_t* desc);
size_t  inr_rams_workspace_bytes(const inr_rams_desc_t* desc, int batch, int height, int width);
int inr_rams_forward(const inr_rams_desc_t* desc, const float* params, const float* x, float* out, int batch,
                     int height, int width, int clip_round, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a-15 (loss side): RAMS shift-tolerant losses (multi-image-super-resolution/utils/loss.py:26-75 l1_loss, :77-127
 * psnr).  y_true, y_pred, mask: [n_images][size][size] fp32.  For every label shift (i, j) in [0, 2*border]^2 the
 * prediction cropped by `border` is compared with the shifted label window under the shifted mask after removing the
 * masked mean brightness difference; out[b] (double) = min over shifts of the masked mean |.| (mode 0, cL1) or
 * max over shifts of 10*log10(65535^2 / masked mean square) (mode 1, cPSNR; the reference then averages over b). */
size_t inr_rams_shift_loss_workspace_bytes(int n_images, int border);
int inr_rams_shift_loss(double* out, const float* y_true, const float* y_pred, const float* mask, int n_images, int size,
                        int border, int mode, void* workspace, size_t workspace_bytes, void* stream);
/* building blocks of the network half of `Trainer.train_step` (utils/training.py:193-209) for the 3x3x3 convolutions
 * 32 -> 32 that carry >= 99 % of RAMS' work (utils/network.py:29-35).  Layouts as in inr_rams_forward: activations
 * [B][D1][D2][D3][32] fp32 (NDHWC), folded kernel w[27 taps][32 cin][32 cout], bias[32]; pad 1 = 'same', 0 = 'valid'.
 *   forward: y = conv(x, w) + bias (+ReLU)                                   -- the inference kernel, on its own
 *   dgrad  : dx = d loss / d x of a 'same' convolution, given dy             -- the same kernel on the flipped, transposed w
 *   wgrad  : gw[27][32][32] = d loss / d w, gb[32] = d loss / d bias (nullable) -- MFMA contraction over the voxels,
 *            fixed-order slab reduction (bitwise reproducible); arithmetic as in the training step: split-fp16 MFMA on
 *            operands staged in LDS (default), f32-input MFMA under inr_debug_set(14, 0); x and dy 16-byte aligned
 * (forward and dgrad here are the f32-input kernels; inside inr_rams_forward / inr_rams_train_* they follow key 14 too) */
int inr_rams_conv3d_forward(float* y, const float* x, const float* w, const float* bias, int B, int D1, int D2, int D3, int pad,
                            int relu, void* stream);
size_t inr_rams_conv3d_dgrad_workspace_bytes(void);
int inr_rams_conv3d_dgrad(float* dx, const float* dy, const float* w, int B, int D1, int D2, int D3, void* workspace,
                          size_t workspace_bytes, void* stream);
size_t inr_rams_conv3d_wgrad_workspace_bytes(int B, int D1, int D2, int D3, int pad);
int inr_rams_conv3d_wgrad(float* gw, float* gb, const float* x, const float* dy, int B, int D1, int D2, int D3, int pad,
                          void* workspace, size_t workspace_bytes, void* stream);

/* the loss half of `Trainer.train_step` (utils/training.py:193-209): loss[b] = cL1 as above and grad_pred[b] =
 * upstream[b] * d loss[b] / d y_pred[b] ([n_images][size][size] fp32, zero on the `border` frame), taken through the
 * best shift as TensorFlow's reduce_min does; upstream nullable (= 1: the gradient of sum_b loss[b], what
 * tape.gradient of the loss vector returns). */
size_t inr_rams_shift_loss_grad_workspace_bytes(int n_images, int border);
int inr_rams_shift_loss_grad(double* loss, float* grad_pred, const float* y_true, const float* y_pred, const float* mask,
                             const float* upstream, int n_images, int size, int border, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- (f)-1: three-compartment hybrid fit (PIA.py:240-283 `three_compartment_fit` / `hybrid_fit`, called at
 * superresHybrid.py:140).  signals: [n_voxels][16] fp64, b-major over b = {0,150,1000,1500} x TE = {0,13,93,143}
 * (the order of PIA.py:263-265).  Runs scipy's bounded trust-region-reflective least squares (what
 * curve_fit(method='trf', maxfev=5000) executes: 2-point Jacobian, x_scale = 1, ftol = xtol = gtol = 1e-8, exact
 * trust-region solver) with the reference's p0 and bounds (PIA.py:269-272), one voxel per lane, fp64.
 * params: [n_voxels][8] = D_ep, D_st, D_lu, T2_ep, T2_st, T2_lu, V_ep, V_st (p0 where the fit exhausts maxfev,
 * PIA.py:276-277); status: scipy termination code (0 = maxfev, 1 gtol, 2 ftol, 3 xtol, 4 both); nfev; cost =
 * half the residual sum of squares at the returned point.  Non-finite input is the caller's to reject
 * (curve_fit(check_finite=True) raises). */
int inr_hybrid_fit(double* params, int* status, int* nfev, double* cost, const double* signals, int64_t n_voxels,
                   void* stream);

/* AutoERD acceptance weights -- the per-pixel outlier rejection master.py runs before a 2-D fit (master.py:77-93).
 * values: [n_pixels][n_acquisitions] fp64 (every acquisition of the slice at that pixel); accept: same shape, fp32, 1 = keep,
 * 0 = reject.  Each pixel's sample is split in two exactly as sklearn.cluster.AgglomerativeClustering(n_clusters=2,
 * affinity='euclidean', linkage='complete') splits it (scipy's nearest-neighbour chain + stable sort: ties fall as they do
 * there); rule 1 (--erd 1, majority voting): a cluster holding >= 2/3 of the acquisitions rejects the other; rule 2 (--erd 2,
 * intensity-cognisant): where erd_map (fp32 [n_pixels], nullable = everywhere) is positive, the cluster with the lower mean
 * is rejected.  2 <= n_acquisitions <= 32. */
int inr_auto_erd(float* accept, const double* values, const float* erd_map, int64_t n_pixels, int n_acquisitions, int rule,
                 void* stream);

/* ---- measurement hooks (bench.py roofline): per-kernel-class HIP-event timing on the launch stream.
 * class ids: 0 = GEMM forward (sine layer), 1 = GEMM input-grad, 2 = GEMM param-grad, 3 = other */
int  inr_prof_enable(int enable);
int  inr_prof_reset(void);
/* synchronises the recorded events; returns launches and total milliseconds for a class */
int  inr_prof_read(int kernel_class, int64_t* launches, double* total_ms);

/* tuning/debug switches (not for production use): key 0 = force the generic GEMM kernel (0/1); key 1 = fp32 MFMA shape
 * of the pipelined GEMM (1 = 16x16x4, default; 0 = 32x32x2); key 2 = hybrid-fit mapping (1 = eight lanes per voxel,
 * default; 0 = one lane per voxel, kept as an independent cross-check); key 3 = GEMM arithmetic of the whole-network
 * entry points (inr_siren_fit / _loss_grad / _forward / _reconstruct): 1 = split-fp16 MFMA (three fp16 products of
 * hi/lo-split operands per fp32 product, fp32 accumulate; default), 0 = f32-input MFMA, 2 = split-fp16 also in the
 * stand-alone layer calls (needs a >= 32 MiB device scratch buffer via inr_debug_set_ptr(1, ptr)); key 5 = serpentine
 * row-tile order between consecutive GEMMs (1 default, 0 off); key 6 = 128 x 256 tiles for the forward GEMMs (1
 * default, 0 = 128 x 128); key 7 = pre-split (HL32) operand path of the fused entry points (1 default); key 10 = its
 * kernel family (2 = persistent with deferred epilogue, default; 1 = persistent, epilogue in line; 0 = one block per
 * tile); key 11 = start stagger between CUs of the persistent kernels (K-steps x 100, 0 default); key 12 = small-network
 * fit (1 = persistent multi-step kernel, default; 0 = two launches per step); key 13 = its rows per block (0 choose,
 * 32, 64); key 14 = RAMS 32->32 convolutions (2 = split-fp16 MFMA, activations staged in LDS, default; 1 = split-fp16,
 * activations from global memory; 0 = f32-input MFMA; +4 forces the LDS-staged kernels); key 15 = which LDS-staged kernel
 * (42 = two blocks of 4 waves x 2 tiles per CU, one staged image each, default; 8 = 8 waves x 1 tile, two images; 4 = 4 waves
 * x 2 tiles, two images; 16 = two-pass 8 waves x 2 tiles); key 16 = last sine layer of a
 * fit step stashes z only (1 default, 0 = act + omega cos); key 17 = poll limit of the small-network kernel's grid barrier (0 = built-in 2^22;
 * tests force the abandon path with 1); key 18 = 64-row tiles for GEMM launches with too few 128-row tiles to fill the chip
 * (1 default); key 19 = cross-layer fused forward for inr_siren_forward / inr_siren_reconstruct (0 default = one launch per layer; 1 =
 * all layers of a 64-row panel in one launch: correct, measured slower, kept as a study -- DESIGN.md); key 20 = the parameter-gradient GEMMs of a fused step as ONE
 * launch behind the input-gradient chain, its row splits chosen for the whole launch (1 default, applied below 200,000 rows, where
 * it pays; 0 = one launch per layer, in line; the gradients' last bits differ between the two: other row ranges per partial sum;
 * inr_launch_count counts every GEMM of the merged launch); key 21 = fewest rows a block of the fused head step takes (16 default;
 * 4 .. 256, multiples of 4: more, smaller blocks at small row counts); key 22 = block count the merged parameter-gradient launch
 * aims at (256 default = one round over the chip; its row splits are this over the layers' tile count); key 23 = rows per block of
 * the fused head step (0 default = the library's rule, at most 128; set it before the workspace of a fit is sized); key 24 = RAMS
 * kernels: 1 = the fusions that keep every bit (training: the data-gradient convolutions apply the ReLU mask / add the residual
 * gradient in their epilogue; inference: long skip in the trunk-closing convolution's epilogue, padded outputs, the stem by rows),
 * 0 = the separate passes they replaced, 2 (default) = also the inference gate of an attention block computed from its first
 * convolution's output so that the second applies it and adds the residual itself (equal up to the rounding of a mean); key 25 = tall slab reductions (more than 128 partial sums per output) in one
 * launch (1, default) or two (0; the same additions in the same order); key 26 = fewest voxels (batch x image) for which RAMS
 * inference takes the gate-ahead form of key 24 = 2 (600,000 default: four 128 x 128 x 9 stacks); key 27 = 1 sends the K-contiguous
 * GEMMs of 512 output columns to the row-owning kernel (a block owns 128 rows x all 512 columns, epilogue in line; 0 = default: measured
 * slower, bit-identical) when they have at least key-28 (default 1024) row panels; key 29 = most 128 x 256 tiles (per 256 CUs; default 192) of a launch that
 * still takes the 64 x 128 tiles of gemm_hp_nt_kernel; key 30 = 0 keeps the head step a kernel of its own (1, default: from key-31 = 768
 * row panels on it rides in the epilogue of the last sine layer, gemm_hp_row_kernel<HPE_HEAD>);
 * keys 8/9 = time-stamp selection of diagnostic builds.
 * PROCESS-GLOBAL and diagnostic only (see "threading" at the top of this file). */
int inr_debug_set(int key, int value);     /* INR_E_INVALID for an unknown key or a value outside the key's range */
int inr_debug_get(int key, int* value);    /* the value a key currently holds */
int inr_debug_reset(void);                 /* every key (and both inr_debug_set_ptr pointers) back to its default */
int inr_debug_set_ptr(int key, void* ptr);   /* key 0: per-wave time-stamp buffer (only honoured by -DINR_STAMPS builds);
                                                 key 1: device scratch for debug key 3 = 2 */

/* Launch-family counters (process-global, monotonic until reset): how many launches each kernel family has received from
 * the host launchers since the last inr_launch_counts_reset().  Lets a parity test assert WHICH family it covered. */
#define INR_LF_HP_PKD      0   /* gemm_hp_pkd_kernel: HL32 operands, persistent grid, epilogue deferred under the next K-loop */
#define INR_LF_HP_PKC      1   /* gemm_hp_pkc_kernel: HL32 operands, persistent grid, epilogue in line */
#define INR_LF_HP_TILE     2   /* gemm_hp_kernel<KC>: HL32 operands, one block per tile */
#define INR_LF_HP_RC       3   /* gemm_hp_kernel<RC>: HL32 parameter gradient */
#define INR_LF_H3          4   /* gemm_h3_kernel: split-fp16, fp32 operands split by the consumer */
#define INR_LF_F32_PIPE16  5   /* gemm_f32_pipe16_kernel: f32-input MFMA 16x16x4 */
#define INR_LF_F32_PIPE    6   /* gemm_f32_pipe_kernel: f32-input MFMA 32x32x2 */
#define INR_LF_F32_GENERIC 7   /* gemm_f32_kernel: guarded generic kernel */
#define INR_LF_SMALL_MULTI 8   /* siren_small_multi_kernel: persistent cooperative small-network fit */
#define INR_LF_SMALL_STEP  9   /* small-network step kernel + reduce/Adam kernel */
#define INR_LF_HP_NARROW   10  /* gemm_hp_nt_kernel: HL32 operands, 64 x 128 tiles (launches too small for the wide tiles) */
#define INR_LF_HP_FUSED_FWD 11 /* siren_fwd_fused_kernel: every sine layer + the head of a forward in ONE launch, panel in LDS */
#define INR_LF_HP_ROW      12  /* gemm_hp_row_kernel: HL32 operands, persistent, a block owns 128 rows x all 512 columns, epilogue in line */
#define INR_LF_COUNT       13
int inr_launch_count(int family, int64_t* count);
int inr_launch_counts_reset(void);

/* diagnostic: s[i] = sin(x[i]), c[i] = cos(x[i]) with the device routine used in the epilogues */
int inr_sincos_probe(float* s, float* c, const float* x, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INRHIP_H */
