// extern "C" surface of libinrhip.so (see include/inrhip.h) + the fused SIREN fit / forward /
// reconstruct orchestration.  Nothing here allocates device memory or synchronises (except
// inr_prof_read and inr_device_caps, which are not on the hot path).
#include <math.h>
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace inr {

// ---- error string -------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiler -------------------------------------------------------------------------------------
struct ProfState {
    std::mutex mu;
    bool enabled = false;
    std::vector<hipEvent_t> pool;                    // recycled events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans[KC_COUNT];
    hipEvent_t open[KC_COUNT] = {nullptr, nullptr, nullptr, nullptr};
    int depth[KC_COUNT] = {0, 0, 0, 0};
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
static ProfState g_prof;
bool prof_enabled() { return g_prof.enabled; }

// ---- launch-family counters (inr_launch_count) ------------------------------------------------------------------------
static std::atomic<long long> g_launches[LF_COUNT];
void count_launch(int family) {
    if (family >= 0 && family < LF_COUNT) g_launches[family].fetch_add(1, std::memory_order_relaxed);
}
// scopes of one class may nest (a launcher calling another launcher): only the outermost pair is recorded
void prof_begin(int kc, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.depth[kc]++ > 0) return;
    hipEvent_t e = g_prof.get();
    (void)hipEventRecord(e, s);
    g_prof.open[kc] = e;
}
void prof_end(int kc, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (--g_prof.depth[kc] > 0) return;
    hipEvent_t e = g_prof.get();
    (void)hipEventRecord(e, s);
    g_prof.spans[kc].push_back({g_prof.open[kc], e});
    g_prof.open[kc] = nullptr;
}

// ---- kernels implemented in gemm_f32.hip / kernels.hip ------------------------------------------------
int gemm_sine_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n,
                      int in_f, int out_f, float omega, hipStream_t stream, const H3Args* h3 = nullptr);
int input_grad_colsum_rows(int64_t n);
int gemm_input_grad(float* dz_prev, const float* dz, const float* W, const float* mul, int64_t n, int in_f,
                    int out_f, float* colsum_slab, int* slab_rows, hipStream_t stream, const H3Args* h3 = nullptr);
size_t h3_planes_bytes(long long weights);
int h3_tensor_amax(unsigned* out, const float* x, long long n, hipStream_t stream, unsigned floor_bits = 0);
int h3_weight_split(const float* const* W, const int* out_f, const int* in_f, int layers, _Float16* planes,
                    unsigned* amax, unsigned* zero_slots, int n_zero, hipStream_t stream);
int param_grad_splits(int64_t n, int in_f, int out_f);
// pre-split (HL32) path: gemm_hp.inc
bool hp_head_ok(int hidden);
bool hp_row_head_ok(int64_t n, int hidden, int in_f);
int hp_row_head_rows(int64_t n);
int hp_sine_forward_head(char* dz_hl, const char* x_hl, const char* W_hl, const float* bias, int64_t n, int in_f, int out_f, float omega,
                         HpScale sa, HpScale sb, HpScale dz_so, const float* head_w, const float* head_b, const float* target,
                         const float* weight, int64_t count_total, float* slab_b, float* slab_w, float* part_loss, float* part_g,
                         unsigned* amax_out, hipStream_t stream);
int gemm_build_flags();
size_t hp_prep_part_bytes();
int hp_weight_prep(const float* const* W, const int* out_f, const int* in_f, int layers, char* planes, unsigned* slots,
                   unsigned* part, float* head_bound, const float* head_W, const float* head_b, int hidden, const unsigned* tmax,
                   const unsigned* wtmax, float inv_count, float omega, hipStream_t stream, const float* const* bias = nullptr,
                   const float* layer_omega = nullptr, float* act_bound = nullptr, const unsigned* x_amax = nullptr);
int hp_convert(char* out, const float* x, long long rows, int cols, HpScale sc, hipStream_t stream);
int hp_unconvert(float* out, const char* x, long long rows, int cols, HpScale sc, hipStream_t stream);
int hp_sine_forward(char* act_hl, float* dact, const char* x_hl, const char* W_hl, const float* bias, int64_t n, int in_f,
                    int out_f, float omega, HpScale sa, HpScale sb, int reverse_m, hipStream_t stream, bool z_only = false,
                    HpScale so = HpScale{});
bool hp_z_stash_ok(int in_f);
bool hp_grid_fourier_ok(int m, int dim);
int hp_grid_fourier_hl(char* x_hl, unsigned* x_amax, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows, const float* B,
                       int m, hipStream_t stream);
bool hp_fused_forward_ok(int in_f, int hidden, int n_sine);
int hp_fused_forward(float* y, const char* x_hl, const unsigned* x_amax, int64_t n, int in_f, int hidden, int n_sine,
                     const char* const* W_hl, const float* const* bias, const unsigned* const* w_amax, float first_omega,
                     float hidden_omega, const float* head_W, const float* head_b, int use_clamp, float clamp_min,
                     hipStream_t stream);
extern tune_int g_hp_zhead;
extern tune_int g_hp_head_rows;
extern tune_int g_rams_epi_fuse;       // key 24 (rams.hip)
extern tune_int g_reduce_onepass;      // key 25 (kernels.hip)
extern tune_int g_rams_pregate_min_vox;   // key 26 (rams.hip)
extern tune_int g_hp_narrow_max_tiles;   // key 29 (gemm_f32.hip)
extern tune_int g_hp_row_head, g_hp_row_head_min_tiles;   // keys 30, 31 (gemm_f32.hip: the head step fused into the last sine layer)
extern tune_int g_hp_row, g_hp_row_min_tiles;   // keys 27, 28 (gemm_f32.hip: the row-owning 128 x 512 kernel)
extern tune_int g_hp_head_min_rows;   // key 21 (gemm_f32.hip)
int hp_input_grad_max_rows(int64_t n);
int hp_input_grad(char* dzprev_hl, const char* dz_hl, const char* WT_hl, const float* mul, int64_t n, int in_f, int out_f,
                  float* colsum_slab, int* colsum_rows, unsigned* amax_out, HpScale sa, HpScale sb, HpScale so,
                  hipStream_t stream);
int64_t hp_head_blocks(int64_t n);
int hp_param_grad_splits(int64_t n, int in_f, int out_f);
int hp_param_grad_slabs(float* slabs, int splits, const char* dz_hl, const char* x_hl, int64_t n, int in_f, int out_f,
                        HpScale sa, HpScale sb, hipStream_t stream);
int hp_param_grad_multi_max();
int hp_param_grad_multi(const HpParamGradJob* jobs, int njobs, int64_t n, hipStream_t stream);
int hp_head_forward(float* y, const char* a_hl, const float* W, const float* bias, int64_t n, int hidden, int use_clamp,
                    float clamp_min, hipStream_t stream, bool from_z = false, float omega = 0.f, HpScale sa = HpScale{});
int hp_head_step(char* dz_hl, float* slab_b, float* slab_w, float* part_loss, float* part_g, const char* a_hl,
                 const float* dact, const float* W, const float* bias, const float* t, const float* wgt, int64_t n, int hidden,
                 int64_t count_total, unsigned* amax_out, HpScale so, hipStream_t stream, bool from_z = false,
                 float omega = 0.f, const float* g_ext = nullptr, HpScale sa = HpScale{});
int hp_head_bound_ext(float* head_bound, const unsigned* gmax, const float* head_W, int hidden, float omega, hipStream_t stream);
int gemm_param_grad_slabs(float* slabs, int splits, const float* dz, const float* x, int64_t n, int in_f,
                          int out_f, hipStream_t stream, const H3Args* h3 = nullptr);
int launch_mgrid(float* out, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows, hipStream_t st);
int launch_fourier(float* out, const float* x, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows,
                   const float* B, int m, hipStream_t st);
int launch_head_forward(float* y, const float* a, const float* W, const float* b, int64_t n, int hidden,
                        int out_f, int use_clamp, float clamp_min, hipStream_t st, float* dy = nullptr);
int gemm_tanh_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n, int in_f,
                      int out_f, float scale, hipStream_t stream);
int mse_blocks(int64_t count);
int launch_mse(float* gy, float* loss, const float* y, const float* t, const float* w, int64_t count,
               float* partial, hipStream_t st, int64_t count_total = 0);
int launch_head_dz(float* dz, const float* gy, const float* W, const float* dact, int64_t n, int hidden,
                   int out_f, hipStream_t st);
int64_t colsum_ws_floats(int64_t n, int C, int G);
int launch_colsum(float* out, const float* X, const float* g, int64_t n, int C, int G, float* slab,
                  hipStream_t st);
int64_t reduce_tmp_floats(int64_t nslabs, int64_t len);
int launch_reduce_slabs(float* out, const float* slab, int nslabs, int64_t len, float* tmp, hipStream_t st);
bool head_fused_ok(int hidden, int out_f, const void* a, const void* b, const void* c, const void* d);
int64_t head_fused_blocks(int64_t n);
int launch_head_bwd_fused(float* dz, float* slab_b, float* slab_w, const float* gy, const float* W, const float* a,
                          const float* dact, int64_t n, int hidden, hipStream_t st, unsigned* amax_out = nullptr);
bool head_step_fused_ok(int hidden, int out_f, const void* a, const void* b, const void* c, const void* d);
int launch_head_step_fused(float* dz, float* slab_b, float* slab_w, float* part_loss, float* part_g, const float* a,
                           const float* dact, const float* W, const float* bias, const float* t, const float* wgt,
                           int64_t n, int hidden, int64_t count_total, hipStream_t st, unsigned* amax_out);
int launch_finish_sum(float* out, const float* partial, int nparts, float scale, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t count, int64_t step, double lr, double b1,
                double b2, double eps, hipStream_t st);
int launch_sincos_probe(float* s, float* c, const float* x, int64_t n, hipStream_t st);
int launch_mul(float* out, const float* a, const float* b, int64_t count, hipStream_t st);
int launch_acquisition_products(float* out, const float* r0, const float* r1, const float* r2, const float* r3, int64_t nvox,
                                int n1, int n2, int n3, hipStream_t st);
int metric_workspace_doubles(int nimg);
int launch_psnr(double* out, const float* x, const float* y, int nimg, int64_t per_image, double data_range,
                double* ws, hipStream_t st);
int launch_ssim(double* out, const float* x, const float* y, int nimg, int H, int W, int win, double data_range,
                int use_mask, float mask_thr, double* ws, hipStream_t st);
int launch_adc(float* out, const float* data, const float* bvals, int64_t npix, int nb, hipStream_t st);
int launch_rescale_linear(float* out, const float* in, int nimg, int H, int W, int OH, int OW, hipStream_t st);
int launch_auto_erd(float* accept, const double* values, const float* erd_map, int64_t npix, int n, int rule, hipStream_t st);
int launch_hybrid_fit(double* params, int* status, int* nfev, double* cost, const double* signals, int64_t n,
                      hipStream_t st);
void set_hybrid_variant(int v);
size_t rams_conv3d_wgrad_ws_floats(long long nvox);
int rams_conv3d_forward(float* y, const float* x, const float* w, const float* bias, int B, int D1, int D2, int D3, int pad,
                        int relu, hipStream_t st);
int rams_conv3d_dgrad_same(float* dx, const float* dy, const float* w, int B, int D1, int D2, int D3, float* ws, hipStream_t st);
int rams_conv3d_wgrad(float* gw, float* gb, const float* x, const float* dy, int B, int D1, int D2, int D3, int pad, float* ws,
                      hipStream_t st);
int rams_conv3d_wgrad_auto(float* gw, float* gb, const float* x, const float* dy, int B, int D1, int D2, int D3, int pad, float* ws,
                      hipStream_t st);
int launch_shift_loss_grad(double* loss, float* grad, const float* y_true, const float* y_pred, const float* mask,
                           const float* upstream, int nimg, int size, int border, double* ws, hipStream_t st);
int launch_shift_loss(double* out, const float* y_true, const float* y_pred, const float* mask, int nimg, int size,
                      int border, int mode, double* ws, hipStream_t st);
long long rams_param_floats(const inr_rams_desc_t* d);
long long rams_train_param_floats(const inr_rams_desc_t* d);
int rams_train_param_offsets(const inr_rams_desc_t* d, int64_t* offsets, int max_layers);
size_t rams_train_workspace_floats(const inr_rams_desc_t* d, int B, int H, int W);
int rams_train_grads(const inr_rams_desc_t* d, const float* raw, float* raw_grad, const float* x, const float* y_true,
                     const float* mask, double* loss, float* pred, int B, int H, int W, float* ws, hipStream_t st);
size_t rams_workspace_floats(const inr_rams_desc_t* d, int B, int H, int W);
int rams_forward_impl(const inr_rams_desc_t* d, const float* params, const float* x, float* out, int B, int H, int W,
                      int clip_round, float* ws, hipStream_t st);
bool small_path_ok(const inr_siren_desc_t* d, int64_t n);
size_t small_workspace_floats(const inr_siren_desc_t* d, int64_t n, long long P);
bool small_multi_ok(const inr_siren_desc_t* d, int64_t n);
size_t small_multi_workspace_floats(const inr_siren_desc_t* d, int64_t n, long long P);
int small_fit_multi(const inr_siren_desc_t* d, const long long* w_off, const long long* b_off, long long P, float* params,
                    float* grads, float* m, float* v, const float* x, const float* targets, const float* weights, int n_acq,
                    int first_acq, int64_t n, int64_t first_step, int n_steps, double lr, double b1, double b2, double eps,
                    float* losses, float* ws, hipStream_t st);
int small_fit_step(const inr_siren_desc_t* d, const long long* w_off, const long long* b_off, long long P, float* params,
                   float* grads, float* m, float* v, const float* x, const float* target, const float* weight, int64_t n,
                   int64_t step, double lr, double b1, double b2, double eps, float* loss_out, float* ws, hipStream_t st);
extern tune_int g_force_generic;
extern tune_int g_small_rows, g_small_spin_limit;
extern tune_int g_rams_h3, g_rams_force_lds, g_rams_lds_waves;
tune_int g_small_multi{1};  // small networks: 1 = persistent multi-step kernel (default), 0 = two launches per step
extern tune_int g_mfma16;
extern tune_int g_h3;
extern tune_int g_h3_wide;
static tune_int g_h3_serpentine{1};
static tune_int g_hp{1};  // pre-split (HL32) GEMM path inside the fused entry points; inr_debug_set(7, 0) falls back to gemm_h3
extern char* g_h3_scratch;
extern unsigned long long* g_stamps;
extern tune_int g_stamp_class, g_stamp_nth;
extern tune_int g_hp_persistent, g_hp_stagger, g_hp_narrow, g_hp_fused_fwd;

// ---- shared helpers ---------------------------------------------------------------------------------
static const int64_t MAX_ROWS = (1ll << 31) - 256;

static int check_desc(const inr_siren_desc_t* d) {
    INR_REQUIRE(d != nullptr, INR_E_INVALID, "siren descriptor is null");
    INR_REQUIRE(d->in_features >= 1 && d->hidden_features >= 1 && d->hidden_layers >= 0 && d->out_features >= 1,
                INR_E_INVALID, "bad siren descriptor (in=%d hidden=%d layers=%d out=%d)", d->in_features,
                d->hidden_features, d->hidden_layers, d->out_features);
    return 0;
}

struct Layout {
    int n_sine;                      // 1 + hidden_layers
    std::vector<int64_t> w_off, b_off;  // per layer, head last
    std::vector<int> fan_in, fan_out;
    int64_t total;
};

static Layout make_layout(const inr_siren_desc_t* d) {
    Layout L;
    L.n_sine = 1 + d->hidden_layers;
    int64_t off = 0;
    for (int l = 0; l <= L.n_sine; ++l) {
        const int fin = (l == 0) ? d->in_features : d->hidden_features;
        const int fout = (l == L.n_sine) ? d->out_features : d->hidden_features;
        L.fan_in.push_back(fin);
        L.fan_out.push_back(fout);
        L.w_off.push_back(off);
        off += (int64_t)round_up((size_t)fin * fout, 4);
        L.b_off.push_back(off);
        off += (int64_t)round_up((size_t)fout, 4);
    }
    L.total = off;
    return L;
}

static size_t max2(size_t a, size_t b) { return a > b ? a : b; }

static size_t param_grad_ws_floats(int64_t n, int in_f, int out_f) {
    const int s1 = param_grad_splits(n, in_f, out_f), s2 = hp_param_grad_splits(n, in_f, out_f);
    const int splits = s1 > s2 ? s1 : s2;
    const size_t len = (size_t)in_f * out_f;
    const size_t slabs = (size_t)splits * len + (size_t)reduce_tmp_floats(splits, (int64_t)len);
    return max2(slabs, (size_t)colsum_ws_floats(n, out_f, 1));
}

// gW = dz^T x (row-split slabs + fixed-order reduce); gb = colsum(dz) when requested
static int param_grad(float* gW, float* gb, const float* dz, const float* x, int64_t n, int in_f, int out_f,
                      float* ws, hipStream_t st, const H3Args* h3 = nullptr) {
    const int splits = param_grad_splits(n, in_f, out_f);
    const int64_t len = (int64_t)in_f * out_f;
    if (int rc = gemm_param_grad_slabs(ws, splits, dz, x, n, in_f, out_f, st, h3)) return rc;
    if (int rc = launch_reduce_slabs(gW, ws, splits, len, ws + (int64_t)splits * len, st)) return rc;
    if (gb) {
        if (int rc = launch_colsum(gb, dz, nullptr, n, out_f, 1, ws, st)) return rc;
    }
    return 0;
}

static size_t input_grad_ws_floats(int64_t n, int in_f) {
    const int64_t rows = input_grad_colsum_rows(n);
    return max2((size_t)(rows * in_f + reduce_tmp_floats(rows, in_f)), (size_t)colsum_ws_floats(n, in_f, 1));
}

// dz_prev = (dz W) * dact_prev; gb_prev (nullable) = colsum(dz_prev), fused into the GEMM epilogue when the
// fast kernel runs, otherwise a separate column-sum pass
static int input_grad(float* dz_prev, float* gb_prev, const float* dz, const float* W, const float* dact_prev,
                      int64_t n, int in_f, int out_f, float* ws, hipStream_t st, const H3Args* h3 = nullptr) {
    int slab_rows = 0;
    float* slab = (gb_prev && dact_prev) ? ws : nullptr;
    if (int rc = gemm_input_grad(dz_prev, dz, W, dact_prev, n, in_f, out_f, slab, &slab_rows, st, h3)) return rc;
    if (!gb_prev) return 0;
    if (slab_rows > 0) return launch_reduce_slabs(gb_prev, ws, slab_rows, in_f, ws + (int64_t)slab_rows * in_f, st);
    return launch_colsum(gb_prev, dz_prev, nullptr, n, in_f, 1, ws, st);
}

// ---- split-fp16 GEMM context of a network (gemm_h3.inc): weight planes + scale slots inside the caller's workspace ----
// slots: [l] = max|W_l|, [8 + l] = max|dz_l| (both rebuilt every step), [24] = max|x|
struct H3Ctx {
    bool on = false;
    _Float16* planes = nullptr;
    unsigned* slots = nullptr;
    unsigned* part = nullptr;           // scratch of the per-step weight statistics (gemm_hp.inc)
    std::vector<long long> plane_off;   // halves, per sine layer
};
static bool h3_eligible(const Layout& L) {
    if (!g_h3 || L.n_sine > 8) return false;
    for (int l = 0; l < L.n_sine; ++l)
        if (L.fan_in[l] % 32 != 0 || L.fan_out[l] % 32 != 0) return false;
    return true;
}
static size_t h3_ctx_bytes(const Layout& L) {
    long long w = 0;
    for (int l = 0; l < L.n_sine; ++l) w += (long long)L.fan_in[l] * L.fan_out[l];
    return round_up(256 + hp_prep_part_bytes() + h3_planes_bytes(w), 256);
}
static H3Ctx h3_make_ctx(const Layout& L, char* region) {
    H3Ctx c;
    c.on = true;
    c.slots = reinterpret_cast<unsigned*>(region);
    c.part = reinterpret_cast<unsigned*>(region + 256);
    c.planes = reinterpret_cast<_Float16*>(region + 256 + hp_prep_part_bytes());
    long long off = 0;
    for (int l = 0; l < L.n_sine; ++l) {
        c.plane_off.push_back(off);
        off += 4ll * L.fan_in[l] * L.fan_out[l];
    }
    return c;
}
// split every sine layer's weights (once per optimizer step / forward call) and zero the dz slots
static int h3_refresh_weights(const H3Ctx& c, const Layout& L, const float* params, hipStream_t st) {
    const float* W[8];
    int of[8], inf[8];
    for (int l = 0; l < L.n_sine; ++l) {
        W[l] = params + L.w_off[l];
        of[l] = L.fan_out[l];
        inf[l] = L.fan_in[l];
    }
    return h3_weight_split(W, of, inf, L.n_sine, c.planes, c.slots, c.slots, 16, st);
}
static H3Args h3_forward_args(const H3Ctx& c, const Layout& L, int l) {
    H3Args a;
    const long long n = (long long)L.fan_in[l] * L.fan_out[l];
    a.a_amax = (l == 0) ? c.slots + 24 : nullptr;   // sine outputs are in [-1, 1]; the network input is whatever it is
    a.b_amax = c.slots + l;
    a.Bh = c.planes + c.plane_off[l];
    a.Bl = a.Bh + n;
    a.reverse_m = g_h3_serpentine ? (l & 1) : 0;      // layer l+1 starts where layer l stopped writing
    return a;
}
static H3Args h3_input_grad_args(const H3Ctx& c, const Layout& L, int l) {
    H3Args a;
    const long long n = (long long)L.fan_in[l] * L.fan_out[l];
    a.a_amax = c.slots + 8 + l;
    a.b_amax = c.slots + l;
    a.Bh = c.planes + c.plane_off[l] + 2 * n;
    a.Bl = a.Bh + n;
    a.amax_out = c.slots + 8 + l - 1;
    // backward chain: head pass (front to back) -> dW_L (back to front) -> dX_L (front to back) -> dW_{L-1} ... : every
    // kernel starts on the rows the one before it touched last
    a.reverse_m = 0;
    return a;
}
static H3Args h3_param_grad_args(const H3Ctx& c, int l) {
    H3Args a;
    a.a_amax = c.slots + 8 + l;
    a.b_amax = (l == 0) ? c.slots + 24 : nullptr;
    a.reverse_m = g_h3_serpentine ? 1 : 0;
    return a;
}


// ---- pre-split (HL32) context: gemm_hp.inc ------------------------------------------------------------------------------
// Shares the H3Ctx region: 256 bytes of slots, the statistics scratch, then 8 bytes per weight (HL32 [out][in], then HL32
// [in][out] per layer).
// slots: [l] max|W_l| bits, [8 + l] measured max|dz_l| bits, [16 + l] wnorm_l (float bits) -- all three rebuilt every step;
// [24] max|x| (floor 1), [25] max|target|, [26] max|weight| (per call); [27] bound of the head's dz (float, every step)
static bool hp_eligible(const inr_siren_desc_t* d, const Layout& L) {
    if (!g_hp || !g_h3 || g_force_generic || !h3_eligible(L)) return false;
    return d->out_features == 1 && hp_head_ok(d->hidden_features);
}
struct HpNet {
    const H3Ctx* c;
    const Layout* L;
    char* w_hl(int l) const { return reinterpret_cast<char*>(c->planes) + 2 * c->plane_off[l]; }
    char* wT_hl(int l) const { return w_hl(l) + 4ll * L->fan_in[l] * L->fan_out[l]; }
    HpScale w_scale(int l) const { HpScale s; s.meas = c->slots + l; s.mul = 1.f; return s; }
    HpScale x_scale() const { HpScale s; s.meas = c->slots + 24; s.mul = 1.f; return s; }
    // input of sine layer l: the network input (measured max|x|) or the output of layer l - 1, whose image is scaled from the
    // a-priori bound of that layer (slots 32 + l - 1, written by the per-step weight preparation: gemm_hp.inc, act_bound)
    HpScale act_scale(int l) const {
        if (l == 0) return x_scale();
        HpScale s;
        s.wn = reinterpret_cast<const float*>(c->slots + 32 + l - 1);
        s.mul = 1.f;
        s.kmax = 40;
        return s;
    }
    float* act_bounds() const { return reinterpret_cast<float*>(c->slots + 32); }
    float* head_bound() const { return reinterpret_cast<float*>(c->slots + 27); }
};
// this step's weights as HL32 images + their scales; with `head` (fit steps) also the a-priori bound of the head's dz
struct HpHeadBoundArgs {
    const float* W;
    const float* b;
    const unsigned* tmax;
    const unsigned* wtmax;
    float inv_count, omega;
};
// x_measured: slot 24 holds max|x| of the rows this step runs on (the fit / forward entry points measure it first); false:
// the input is a coordinate grid or its Fourier features, |x| <= 1 (the dense re-sampling prepares the weights once per call,
// before any chunk's input exists)
static int hp_refresh_weights(const HpNet& net, const inr_siren_desc_t* d, const float* params, hipStream_t st, bool x_measured,
                              const HpHeadBoundArgs* head = nullptr) {
    const Layout& L = *net.L;
    const float *W[8], *bias[8];
    float om[8];
    int of[8], inf[8];
    for (int l = 0; l < L.n_sine; ++l) {
        W[l] = params + L.w_off[l];
        bias[l] = params + L.b_off[l];
        om[l] = l == 0 ? d->first_omega : d->hidden_omega;
        of[l] = L.fan_out[l];
        inf[l] = L.fan_in[l];
    }
    return hp_weight_prep(W, of, inf, L.n_sine, reinterpret_cast<char*>(net.c->planes), net.c->slots, net.c->part,
                          head ? net.head_bound() : nullptr, head ? head->W : nullptr, head ? head->b : nullptr,
                          L.fan_in[L.n_sine], head ? head->tmax : nullptr, head ? head->wtmax : nullptr,
                          head ? head->inv_count : 0.f, head ? head->omega : 0.f, st, bias, om, net.act_bounds(),
                          x_measured ? net.c->slots + 24 : nullptr);
}

static size_t head_backward_ws_floats(int64_t n, int hidden, int out_f) {
    const int64_t blocks = head_fused_blocks(n);
    const size_t fused = (size_t)(2 * blocks * hidden + reduce_tmp_floats(blocks, hidden) + 2 * blocks);   // + loss / sum-g partials
    const size_t a = (size_t)colsum_ws_floats(n, hidden, out_f);
    const size_t b = (size_t)colsum_ws_floats(n, out_f, 1);
    return max2(fused, max2(a, b));
}

// head backward: dz_last = (gy W) * dact_last, gW/gb of the head, and (optionally) gb_last = colsum(dz_last),
// the bias gradient of the last sine layer.  One fused pass when out_features == 1.
static int head_backward(float* dz_last, float* gW, float* gb, float* gb_last, const float* gy, const float* a_last,
                         const float* dact_last, const float* W, int64_t n, int hidden, int out_f, float* ws,
                         hipStream_t st, unsigned* dz_amax = nullptr) {
    if (dz_last && gW && dact_last && head_fused_ok(hidden, out_f, dz_last, a_last, dact_last, W)) {
        const int64_t blocks = head_fused_blocks(n);
        float* slab_b = ws;
        float* slab_w = ws + blocks * hidden;
        float* tmp = ws + 2 * blocks * hidden;
        if (int rc = launch_head_bwd_fused(dz_last, slab_b, slab_w, gy, W, a_last, dact_last, n, hidden, st, dz_amax))
            return rc;
        if (gb_last) {
            if (int rc = launch_reduce_slabs(gb_last, slab_b, (int)blocks, hidden, tmp, st)) return rc;
        }
        if (int rc = launch_reduce_slabs(gW, slab_w, (int)blocks, hidden, tmp, st)) return rc;
        if (gb) return launch_colsum(gb, gy, nullptr, n, out_f, 1, ws, st);
        return 0;
    }
    if (gW) {
        if (int rc = launch_colsum(gW, a_last, gy, n, hidden, out_f, ws, st)) return rc;
    }
    if (gb) {
        if (int rc = launch_colsum(gb, gy, nullptr, n, out_f, 1, ws, st)) return rc;
    }
    if (dz_last) {
        if (int rc = launch_head_dz(dz_last, gy, W, dact_last, n, hidden, out_f, st)) return rc;
        if (dz_amax) {
            if (int rc = h3_tensor_amax(dz_amax, dz_last, (long long)n * hidden, st)) return rc;
        }
        if (gb_last) return launch_colsum(gb_last, dz_last, nullptr, n, hidden, 1, ws, st);
    }
    return 0;
}

}  // namespace inr

using namespace inr;

extern "C" {

int inr_version(void) { return INR_ABI_VERSION; }
int inr_build_flags(void) { return gemm_build_flags(); }
const char* inr_last_error(void) { return g_err; }

int inr_device_caps(int device, inr_device_caps_t* out) {
    INR_REQUIRE(out != nullptr, INR_E_INVALID, "caps output is null");
    hipDeviceProp_t prop;
    INR_HIP(hipGetDeviceProperties(&prop, device));
    memset(out, 0, sizeof(*out));
    out->abi_version = INR_ABI_VERSION;
    out->device = device;
    out->compute_units = prop.multiProcessorCount;
    out->wavefront_size = prop.warpSize;
    out->lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    out->clock_khz = prop.clockRate;
    out->hbm_bytes = (int64_t)prop.totalGlobalMem;
    strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    return 0;
}

int inr_mgrid(float* out, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows, void* stream) {
    INR_REQUIRE(out && shape, INR_E_INVALID, "inr_mgrid: null pointer");
    INR_REQUIRE(row_begin >= 0 && n_rows >= 0, INR_E_INVALID, "inr_mgrid: negative row range");
    return launch_mgrid(out, shape, dim, row_begin, n_rows, (hipStream_t)stream);
}

int inr_fourier_map(float* out, const float* x, const float* B, int64_t n, int d, int m, void* stream) {
    INR_REQUIRE(out && x && B, INR_E_INVALID, "inr_fourier_map: null pointer");
    INR_REQUIRE(n >= 0 && d >= 1 && m >= 1, INR_E_INVALID, "inr_fourier_map: bad sizes n=%lld d=%d m=%d",
                (long long)n, d, m);
    return launch_fourier(out, x, nullptr, d, 0, n, B, m, (hipStream_t)stream);
}

int inr_grid_fourier_map(float* out, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows,
                         const float* B, int m, void* stream) {
    INR_REQUIRE(out && shape && B, INR_E_INVALID, "inr_grid_fourier_map: null pointer");
    INR_REQUIRE(row_begin >= 0 && n_rows >= 0 && m >= 1, INR_E_INVALID, "inr_grid_fourier_map: bad sizes");
    return launch_fourier(out, nullptr, shape, dim, row_begin, n_rows, B, m, (hipStream_t)stream);
}

int inr_sine_layer_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n,
                           int in_features, int out_features, float omega, void* stream) {
    INR_REQUIRE(act && x && W, INR_E_INVALID, "inr_sine_layer_forward: null pointer");
    INR_REQUIRE(n >= 0 && n <= MAX_ROWS && in_features >= 1 && out_features >= 1, INR_E_INVALID,
                "inr_sine_layer_forward: bad sizes n=%lld in=%d out=%d", (long long)n, in_features, out_features);
    if (n == 0) return 0;
    return gemm_sine_forward(act, dact, x, W, b, n, in_features, out_features, omega, (hipStream_t)stream);
}

int inr_tanh_layer_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n,
                           int in_features, int out_features, float scale, void* stream) {
    INR_REQUIRE(act && x && W, INR_E_INVALID, "inr_tanh_layer_forward: null pointer");
    INR_REQUIRE(n >= 0 && n <= MAX_ROWS && in_features >= 1 && out_features >= 1, INR_E_INVALID,
                "inr_tanh_layer_forward: bad sizes");
    if (n == 0) return 0;
    return gemm_tanh_forward(act, dact, x, W, b, n, in_features, out_features, scale, (hipStream_t)stream);
}

int inr_linear_tanh_head_forward(float* y, float* dy, const float* a, const float* W, const float* b, int64_t n,
                                 int in_features, int out_features, float scale, void* stream) {
    INR_REQUIRE(y && a && W, INR_E_INVALID, "inr_linear_tanh_head_forward: null pointer");
    INR_REQUIRE(n >= 0 && in_features >= 1 && out_features >= 1, INR_E_INVALID, "inr_linear_tanh_head_forward: bad sizes");
    return launch_head_forward(y, a, W, b, n, in_features, out_features, 2, scale, (hipStream_t)stream, dy);
}

int inr_acquisition_products(float* out, const float* raw_b0, const float* raw_b1, const float* raw_b2, const float* raw_b3,
                             int64_t n_voxels, int n1, int n2, int n3, void* stream) {
    INR_REQUIRE(out && raw_b0 && raw_b1 && raw_b2 && raw_b3, INR_E_INVALID, "inr_acquisition_products: null pointer");
    INR_REQUIRE(n_voxels >= 0 && n1 >= 1 && n2 >= 1 && n3 >= 1 && (long long)n1 * n2 * n3 < (1 << 24), INR_E_INVALID,
                "inr_acquisition_products: bad sizes");
    return launch_acquisition_products(out, raw_b0, raw_b1, raw_b2, raw_b3, n_voxels, n1, n2, n3, (hipStream_t)stream);
}

int inr_mul(float* out, const float* a, const float* b, int64_t count, void* stream) {
    INR_REQUIRE(out && a && b && count >= 0, INR_E_INVALID, "inr_mul: bad arguments");
    return launch_mul(out, a, b, count, (hipStream_t)stream);
}

int inr_linear_head_forward(float* y, const float* a, const float* W, const float* b, int64_t n, int in_features,
                            int out_features, int use_clamp, float clamp_min, void* stream) {
    INR_REQUIRE(y && a && W, INR_E_INVALID, "inr_linear_head_forward: null pointer");
    INR_REQUIRE(n >= 0 && in_features >= 1 && out_features >= 1, INR_E_INVALID, "inr_linear_head_forward: bad sizes");
    return launch_head_forward(y, a, W, b, n, in_features, out_features, use_clamp, clamp_min, (hipStream_t)stream);
}

size_t inr_mse_workspace_bytes(int64_t count) { return (size_t)mse_blocks(count > 0 ? count : 1) * sizeof(float); }

int inr_mse_loss_grad(float* gy, float* loss, const float* y, const float* t, const float* w, int64_t count,
                      void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(gy && loss && y && t, INR_E_INVALID, "inr_mse_loss_grad: null pointer");
    INR_REQUIRE(count >= 1, INR_E_INVALID, "inr_mse_loss_grad: count must be >= 1");
    INR_REQUIRE(workspace && workspace_bytes >= inr_mse_workspace_bytes(count), INR_E_WORKSPACE,
                "inr_mse_loss_grad: workspace too small");
    return launch_mse(gy, loss, y, t, w, count, (float*)workspace, (hipStream_t)stream);
}

size_t inr_head_backward_workspace_bytes(int64_t n, int hidden, int out_features) {
    return head_backward_ws_floats(n > 0 ? n : 1, hidden, out_features) * sizeof(float);
}

int inr_linear_head_backward(float* dz_last, float* gW, float* gb, float* gb_last, const float* gy,
                             const float* a_last, const float* dact_last, const float* W, int64_t n, int hidden,
                             int out_features, void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(gy && a_last && W, INR_E_INVALID, "inr_linear_head_backward: null pointer");
    INR_REQUIRE(n >= 1 && hidden >= 1 && out_features >= 1, INR_E_INVALID, "inr_linear_head_backward: bad sizes");
    INR_REQUIRE(workspace && workspace_bytes >= inr_head_backward_workspace_bytes(n, hidden, out_features),
                INR_E_WORKSPACE, "inr_linear_head_backward: workspace too small");
    INR_REQUIRE(!gb_last || dz_last, INR_E_INVALID, "inr_linear_head_backward: gb_last needs dz_last");
    return head_backward(dz_last, gW, gb, gb_last, gy, a_last, dact_last, W, n, hidden, out_features,
                         (float*)workspace, (hipStream_t)stream);
}

size_t inr_sine_layer_backward_input_workspace_bytes(int64_t n, int in_features) {
    return input_grad_ws_floats(n > 0 ? n : 1, in_features) * sizeof(float);
}

int inr_sine_layer_backward_input(float* dz_prev, float* gb_prev, const float* dz, const float* W,
                                  const float* dact_prev, int64_t n, int in_features, int out_features,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(dz_prev && dz && W, INR_E_INVALID, "inr_sine_layer_backward_input: null pointer");
    INR_REQUIRE(n >= 0 && n <= MAX_ROWS && in_features >= 1 && out_features >= 1, INR_E_INVALID,
                "inr_sine_layer_backward_input: bad sizes");
    if (n == 0) return 0;
    if (gb_prev)
        INR_REQUIRE(workspace && workspace_bytes >= inr_sine_layer_backward_input_workspace_bytes(n, in_features),
                    INR_E_WORKSPACE, "inr_sine_layer_backward_input: workspace too small for gb_prev");
    return input_grad(dz_prev, gb_prev, dz, W, dact_prev, n, in_features, out_features, (float*)workspace,
                      (hipStream_t)stream);
}

size_t inr_linear_param_grad_workspace_bytes(int64_t n, int in_features, int out_features) {
    return param_grad_ws_floats(n > 0 ? n : 1, in_features, out_features) * sizeof(float);
}

int inr_linear_param_grad(float* gW, float* gb, const float* dz, const float* x, int64_t n, int in_features,
                          int out_features, void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(gW && dz && x, INR_E_INVALID, "inr_linear_param_grad: null pointer");
    INR_REQUIRE(n >= 1 && n <= MAX_ROWS && in_features >= 1 && out_features >= 1, INR_E_INVALID,
                "inr_linear_param_grad: bad sizes");
    INR_REQUIRE(workspace && workspace_bytes >= inr_linear_param_grad_workspace_bytes(n, in_features, out_features),
                INR_E_WORKSPACE, "inr_linear_param_grad: workspace too small");
    return param_grad(gW, gb, dz, x, n, in_features, out_features, (float*)workspace, (hipStream_t)stream);
}

int inr_adam_step(float* p, const float* g, float* m, float* v, int64_t count, int64_t step, double lr,
                  double beta1, double beta2, double eps, void* stream) {
    INR_REQUIRE(p && g && m && v, INR_E_INVALID, "inr_adam_step: null pointer");
    INR_REQUIRE(count >= 0 && step >= 1, INR_E_INVALID, "inr_adam_step: count >= 0 and step >= 1 required");
    return launch_adam(p, g, m, v, count, step, lr, beta1, beta2, eps, (hipStream_t)stream);
}

// ---- fused SIREN -------------------------------------------------------------------------------------
int64_t inr_siren_param_count(const inr_siren_desc_t* desc) {
    if (check_desc(desc)) return INR_E_INVALID;
    return make_layout(desc).total;
}

int inr_siren_param_offsets(const inr_siren_desc_t* desc, int64_t* offsets) {
    if (int rc = check_desc(desc)) return rc;
    INR_REQUIRE(offsets != nullptr, INR_E_INVALID, "offsets is null");
    const Layout L = make_layout(desc);
    for (int l = 0; l <= L.n_sine; ++l) {
        offsets[2 * l] = L.w_off[l];
        offsets[2 * l + 1] = L.b_off[l];
    }
    return 0;
}

size_t inr_siren_forward_workspace_bytes(const inr_siren_desc_t* desc, int64_t n) {
    if (check_desc(desc)) return 0;
    return 2 * round_up((size_t)(n > 0 ? n : 1) * desc->hidden_features * sizeof(float), 256) +
           h3_ctx_bytes(make_layout(desc)) + round_up((size_t)(n > 0 ? n : 1) * desc->in_features * sizeof(float), 256);
}

static int siren_forward_impl(const inr_siren_desc_t* d, const Layout& L, const float* params, const float* x,
                              int64_t n, float* y, int use_clamp, float clamp_min, float* buf0, float* buf1,
                              hipStream_t st, const H3Ctx* h3 = nullptr, char* xhl = nullptr, bool xhl_ready = false,
                              bool amax_ready = false) {
    // xhl_ready: the caller has already written the HL32 image of the input and its scale slot (hp_grid_fourier_hl); x unused
    // amax_ready: the caller has measured max|x| into slot 24 (inr_siren_forward: before the weight preparation, which needs it)
    const float* cur = x;
    float* bufs[2] = {buf0, buf1};
    if (h3 && h3->on && !xhl_ready && !amax_ready) {
        if (int rc = h3_tensor_amax(h3->slots + 24, x, (long long)n * L.fan_in[0], st, 0x3f800000u)) return rc;
    }
    if (h3 && h3->on && xhl) {   // pre-split path: every activation lives in HBM as HL32 (gemm_hp.inc)
        const HpNet net{h3, &L};
        if (!xhl_ready) {
            if (int rc = hp_convert(xhl, x, n, L.fan_in[0], net.x_scale(), st)) return rc;
        }
        if (hp_fused_forward_ok(L.fan_in[0], d->hidden_features, L.n_sine)) {
            // every sine layer and the head in ONE launch, the activations of a 64-row panel never leaving LDS (gemm_hp_fwd.inc)
            const char* W[8];
            const float* bias[8];
            const unsigned* wmax[8];
            for (int l = 0; l < L.n_sine; ++l) {
                W[l] = net.w_hl(l);
                bias[l] = params + L.b_off[l];
                wmax[l] = h3->slots + l;
            }
            return hp_fused_forward(y, xhl, h3->slots + 24, n, L.fan_in[0], d->hidden_features, L.n_sine, W, bias, wmax,
                                    d->first_omega, d->hidden_omega, params + L.w_off[L.n_sine], params + L.b_off[L.n_sine],
                                    use_clamp, clamp_min, st);
        }
        const char* in = xhl;
        for (int l = 0; l < L.n_sine; ++l) {
            char* dst = reinterpret_cast<char*>(bufs[l & 1]);
            const float omega = (l == 0) ? d->first_omega : d->hidden_omega;
            if (int rc = hp_sine_forward(dst, nullptr, in, net.w_hl(l), params + L.b_off[l], n, L.fan_in[l], L.fan_out[l],
                                         omega, net.act_scale(l), net.w_scale(l), 0, st, false, net.act_scale(l + 1)))
                return rc;
            in = dst;
        }
        return hp_head_forward(y, in, params + L.w_off[L.n_sine], params + L.b_off[L.n_sine], n, d->hidden_features,
                               use_clamp, clamp_min, st, false, 0.f, net.act_scale(L.n_sine));
    }
    for (int l = 0; l < L.n_sine; ++l) {
        float* dst = bufs[l & 1];
        const float omega = (l == 0) ? d->first_omega : d->hidden_omega;
        H3Args ha;
        if (h3 && h3->on) ha = h3_forward_args(*h3, L, l);
        if (int rc = gemm_sine_forward(dst, nullptr, cur, params + L.w_off[l], params + L.b_off[l], n, L.fan_in[l],
                                       L.fan_out[l], omega, st, (h3 && h3->on) ? &ha : nullptr))
            return rc;
        cur = dst;
    }
    return launch_head_forward(y, cur, params + L.w_off[L.n_sine], params + L.b_off[L.n_sine], n,
                               d->hidden_features, d->out_features, use_clamp, clamp_min, st);
}

int inr_siren_forward(const inr_siren_desc_t* desc, const float* params, const float* x, int64_t n, float* y,
                      int use_clamp, float clamp_min, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(desc)) return rc;
    INR_REQUIRE(params && x && y, INR_E_INVALID, "inr_siren_forward: null pointer");
    INR_REQUIRE(n >= 0 && n <= MAX_ROWS, INR_E_INVALID, "inr_siren_forward: bad row count %lld", (long long)n);
    if (n == 0) return 0;
    INR_REQUIRE(workspace && workspace_bytes >= inr_siren_forward_workspace_bytes(desc, n), INR_E_WORKSPACE,
                "inr_siren_forward: workspace too small");
    INR_REQUIRE(aligned16(workspace), INR_E_ALIGN, "inr_siren_forward: workspace must be 16-byte aligned");
    const Layout L = make_layout(desc);
    const size_t half = round_up((size_t)n * desc->hidden_features * sizeof(float), 256);
    float* b0 = (float*)workspace;
    float* b1 = (float*)((char*)workspace + half);
    bool amax_ready = false;
    H3Ctx h3;
    char* xhl = nullptr;
    if (h3_eligible(L)) {
        h3 = h3_make_ctx(L, (char*)workspace + 2 * half);
        if (hp_eligible(desc, L)) {
            xhl = (char*)workspace + 2 * half + h3_ctx_bytes(L);
            // max|x| first: the a-priori bounds of the layers' outputs (the scales of their images) start from it
            if (int rc = h3_tensor_amax(h3.slots + 24, x, (long long)n * L.fan_in[0], (hipStream_t)stream, 0x3f800000u)) return rc;
            amax_ready = true;
            if (int rc = hp_refresh_weights(HpNet{&h3, &L}, desc, params, (hipStream_t)stream, true)) return rc;
        } else if (int rc = h3_refresh_weights(h3, L, params, (hipStream_t)stream)) {
            return rc;
        }
    }
    return siren_forward_impl(desc, L, params, x, n, y, use_clamp, clamp_min, b0, b1, (hipStream_t)stream, &h3, xhl, false,
                              amax_ready);
}

size_t inr_siren_reconstruct_workspace_bytes(const inr_siren_desc_t* desc, int64_t chunk_rows) {
    if (check_desc(desc) || chunk_rows < 1) return 0;
    const size_t feats = round_up((size_t)chunk_rows * desc->in_features * sizeof(float), 256);
    const size_t act = round_up((size_t)chunk_rows * desc->hidden_features * sizeof(float), 256);
    return 2 * feats + 2 * act + h3_ctx_bytes(make_layout(desc));   // second feature buffer: its HL32 image
}

int inr_siren_reconstruct(const inr_siren_desc_t* desc, const float* params, const int64_t* shape, int dim,
                          const float* B, int m, float* y, int use_clamp, float clamp_min, int64_t chunk_rows,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(desc)) return rc;
    INR_REQUIRE(params && shape && y, INR_E_INVALID, "inr_siren_reconstruct: null pointer");
    INR_REQUIRE(dim >= 1 && dim <= 8, INR_E_INVALID, "inr_siren_reconstruct: dim must be 1..8");
    INR_REQUIRE(chunk_rows >= 1 && chunk_rows <= MAX_ROWS, INR_E_INVALID, "inr_siren_reconstruct: bad chunk_rows");
    if (B)
        INR_REQUIRE(2 * m == desc->in_features, INR_E_INVALID,
                    "inr_siren_reconstruct: in_features (%d) must equal 2*m (%d)", desc->in_features, 2 * m);
    else
        INR_REQUIRE(dim == desc->in_features, INR_E_INVALID,
                    "inr_siren_reconstruct: without B the grid dim (%d) must equal in_features (%d)", dim,
                    desc->in_features);
    INR_REQUIRE(workspace && workspace_bytes >= inr_siren_reconstruct_workspace_bytes(desc, chunk_rows),
                INR_E_WORKSPACE, "inr_siren_reconstruct: workspace too small");
    INR_REQUIRE(aligned16(workspace), INR_E_ALIGN, "inr_siren_reconstruct: workspace must be 16-byte aligned");
    int64_t total = 1;
    for (int a = 0; a < dim; ++a) {
        INR_REQUIRE(shape[a] >= 1, INR_E_INVALID, "inr_siren_reconstruct: shape[%d] must be >= 1", a);
        total *= shape[a];
    }
    const Layout L = make_layout(desc);
    const size_t feats_b = round_up((size_t)chunk_rows * desc->in_features * sizeof(float), 256);
    const size_t act_b = round_up((size_t)chunk_rows * desc->hidden_features * sizeof(float), 256);
    float* feats = (float*)workspace;
    float* b0 = (float*)((char*)workspace + feats_b);
    float* b1 = (float*)((char*)workspace + feats_b + act_b);
    hipStream_t st = (hipStream_t)stream;
    H3Ctx h3;
    char* xhl = nullptr;
    if (h3_eligible(L)) {
        h3 = h3_make_ctx(L, (char*)workspace + feats_b + 2 * act_b);
        if (hp_eligible(desc, L)) {
            xhl = (char*)workspace + feats_b + 2 * act_b + h3_ctx_bytes(L);
            if (int rc = hp_refresh_weights(HpNet{&h3, &L}, desc, params, st, false)) return rc;
        } else if (int rc = h3_refresh_weights(h3, L, params, st)) {
            return rc;
        }
    }
    for (int64_t r0 = 0; r0 < total; r0 += chunk_rows) {
        const int64_t rows = (total - r0 < chunk_rows) ? (total - r0) : chunk_rows;
        // pre-split path with Fourier features: grid -> features -> HL32 image in ONE kernel (no fp32 feature matrix at all)
        const bool direct = xhl && B && hp_grid_fourier_ok(m, dim);
        int rc = direct ? hp_grid_fourier_hl(xhl, h3.slots + 24, shape, dim, r0, rows, B, m, st)
                 : B    ? launch_fourier(feats, nullptr, shape, dim, r0, rows, B, m, st)
                        : launch_mgrid(feats, shape, dim, r0, rows, st);
        if (rc) return rc;
        rc = siren_forward_impl(desc, L, params, feats, rows, y + r0 * desc->out_features, use_clamp, clamp_min, b0,
                                b1, st, &h3, xhl, direct);
        if (rc) return rc;
    }
    return 0;
}

// ---- where the producers of the pre-split step leave their slab rows until launch_finalize sums them (common.h) ----------
// Tensor order = flat parameter order: W_0, b_0, ..., W_{S-1}, b_{S-1}, W_head, b_head.  Offsets in floats from the slab region.
struct HpSlabPlan {
    struct Seg {
        size_t slab, stage1;
        long long len;
        int max_rows;
    };
    std::vector<Seg> seg;       // 2 (S + 1)
    size_t part_loss = 0;       // [head blocks]
    size_t loss_sink = 0;       // where the loss goes when the caller does not want it
    size_t total = 0;
};
static HpSlabPlan hp_slab_plan(const Layout& L, int64_t n) {
    HpSlabPlan p;
    const int S = L.n_sine, H = L.fan_in[S];
    // rows of the head's slabs: one per block of hp_head_step_kernel, or two per 128-row panel when the head step rides in the last sine
    // layer's epilogue (hp_sine_forward_head) -- the plan reserves for whichever is more
    int64_t hb = hp_head_blocks(n);
    if (hp_row_head_rows(n) > hb) hb = hp_row_head_rows(n);
    size_t off = 0;
    auto add = [&](long long len, int rows) {
        HpSlabPlan::Seg sg;
        sg.len = len;
        sg.max_rows = rows;
        sg.slab = off;
        off += round_up((size_t)rows * (size_t)len, 4);
        sg.stage1 = off;
        if (rows > FIN_TALL) off += round_up((size_t)((rows + FIN_GROUP - 1) / FIN_GROUP) * (size_t)len, 4);
        p.seg.push_back(sg);
    };
    for (int l = 0; l < S; ++l) {
        add((long long)L.fan_in[l] * L.fan_out[l], hp_param_grad_splits(n, L.fan_in[l], L.fan_out[l]));
        add(L.fan_out[l], l == S - 1 ? (int)hb : hp_input_grad_max_rows(n));
    }
    add(H, (int)hb);     // W_head: the head step's slab_w
    add(1, (int)hb);     // b_head: its per-block sums of g
    p.part_loss = off;
    off += round_up((size_t)hb, 4);
    p.loss_sink = off;
    off += 4;
    p.total = off;
    return p;
}

// workspace carve for the fit: acts (n_sine x n x H), dacts (n_sine x n x H), y, gy, scratch
struct FitCarve {
    size_t act_b, out_b, scratch_b, total, h3_off, xhl_off;
};
static FitCarve fit_carve(const inr_siren_desc_t* d, const Layout& L, int64_t n) {
    FitCarve c;
    c.act_b = round_up((size_t)n * d->hidden_features * sizeof(float), 256);
    c.out_b = round_up((size_t)n * d->out_features * sizeof(float), 256);
    size_t scratch = head_backward_ws_floats(n, d->hidden_features, d->out_features);
    for (int l = 0; l < L.n_sine; ++l) {
        scratch = max2(scratch, param_grad_ws_floats(n, L.fan_in[l], L.fan_out[l]));
        if (l > 0) scratch = max2(scratch, input_grad_ws_floats(n, L.fan_in[l]));
    }
    const size_t mse = (size_t)mse_blocks((int64_t)n * d->out_features) + 1;
    if (mse > scratch) scratch = mse;
    if (L.n_sine <= 8 && d->out_features == 1) scratch = max2(scratch, hp_slab_plan(L, n).total);   // deferred slabs of the HL32 step
    c.scratch_b = round_up(scratch * sizeof(float), 256);
    c.h3_off = 2 * (size_t)L.n_sine * c.act_b + 2 * c.out_b + c.scratch_b;
    c.xhl_off = c.h3_off + h3_ctx_bytes(L);
    c.total = c.xhl_off + round_up((size_t)n * d->in_features * sizeof(float), 256);   // HL32 image of the network input
    if (small_path_ok(d, n)) {   // the fused small-network step carves the same workspace differently
        const size_t small = round_up(small_workspace_floats(d, n, L.total) * sizeof(float), 256);
        if (small > c.total) c.total = small;
        if (small_multi_ok(d, n)) {
            const size_t multi = round_up(small_multi_workspace_floats(d, n, L.total) * sizeof(float), 256);
            if (multi > c.total) c.total = multi;
        }
    }
    return c;
}

// one forward (with stash) + loss + backward of the layer-wise path; grads land in the flat gradient buffer.
// count_total > 0: this is one row shard of a split fit (mean taken over count_total elements).
static int fit_forward_backward(const inr_siren_desc_t* d, const Layout& L, const float* params, float* grads,
                                std::vector<float*>& act, std::vector<float*>& dact, float* y, float* gy, float* scratch,
                                const float* target, const float* weight, int64_t n, int64_t count_total,
                                float* loss_dst, hipStream_t st, const H3Ctx* h3 = nullptr) {
    const int H = d->hidden_features, O = d->out_features, head = L.n_sine;
    const bool split = h3 && h3->on;
    if (split) {   // this step's weights as fp16 planes; dz scale slots back to zero
        if (int rc = h3_refresh_weights(*h3, L, params, st)) return rc;
    }
    // forward with stash (SRDWI.py:58-59 per layer; dact = omega*cos(.) replaces autograd's saved z)
    for (int l = 0; l < L.n_sine; ++l) {
        const float omega = (l == 0) ? d->first_omega : d->hidden_omega;
        H3Args ha;
        if (split) ha = h3_forward_args(*h3, L, l);
        if (int rc = gemm_sine_forward(act[l + 1], dact[l], act[l], params + L.w_off[l], params + L.b_off[l], n,
                                       L.fan_in[l], L.fan_out[l], omega, st, split ? &ha : nullptr))
            return rc;
    }
    // head forward, loss + dL/dy (superresDWI.py:135), head backward; then the sine layers from last to first, dz
    // overwriting dact in place (bias gradients ride along: the head pass yields gb of the last sine layer, every
    // input-grad GEMM yields gb of the layer below from its epilogue)
    if (head_step_fused_ok(H, O, act[head], dact[head - 1], params + L.w_off[head], scratch)) {
        // one pass over act_L / dact_L does all three (kernels.hip: head_step_fused_kernel)
        const int64_t blocks = head_fused_blocks(n);
        float* slab_b = scratch;
        float* slab_w = scratch + blocks * H;
        float* tmp = scratch + 2 * blocks * H;
        float* part_loss = tmp + reduce_tmp_floats(blocks, H);
        float* part_g = part_loss + blocks;
        if (int rc = launch_head_step_fused(dact[head - 1], slab_b, slab_w, part_loss, part_g, act[head], dact[head - 1],
                                            params + L.w_off[head], params + L.b_off[head], target, weight, n, H,
                                            count_total, st, split ? h3->slots + 8 + head - 1 : nullptr))
            return rc;
        if (int rc = launch_reduce_slabs(grads + L.b_off[head - 1], slab_b, (int)blocks, H, tmp, st)) return rc;
        if (int rc = launch_reduce_slabs(grads + L.w_off[head], slab_w, (int)blocks, H, tmp, st)) return rc;
        const float inv = (float)(1.0 / (double)(count_total > 0 ? count_total : n));
        if (int rc = launch_finish_sum(loss_dst, part_loss, (int)blocks, inv, st)) return rc;
        if (int rc = launch_finish_sum(grads + L.b_off[head], part_g, (int)blocks, 1.0f, st)) return rc;
    } else {
        if (int rc = launch_head_forward(y, act[head], params + L.w_off[head], params + L.b_off[head], n, H, O, 0,
                                         0.f, st))
            return rc;
        if (int rc = launch_mse(gy, loss_dst, y, target, weight, n * O, scratch + 1, st, count_total)) return rc;
        if (int rc = head_backward(dact[head - 1], grads + L.w_off[head], grads + L.b_off[head],
                                   grads + L.b_off[head - 1], gy, act[head], dact[head - 1], params + L.w_off[head], n,
                                   H, O, scratch, st, split ? h3->slots + 8 + head - 1 : nullptr))
            return rc;
    }
    for (int l = L.n_sine - 1; l >= 0; --l) {
        H3Args hp, hi;
        if (split) {
            hp = h3_param_grad_args(*h3, l);
            if (l > 0) hi = h3_input_grad_args(*h3, L, l);
        }
        if (int rc = param_grad(grads + L.w_off[l], nullptr, dact[l], act[l], n, L.fan_in[l], L.fan_out[l],
                                scratch, st, split ? &hp : nullptr))
            return rc;
        if (l > 0) {
            if (int rc = input_grad(dact[l - 1], grads + L.b_off[l - 1], dact[l], params + L.w_off[l], dact[l - 1],
                                    n, L.fan_in[l], L.fan_out[l], scratch, st, split ? &hi : nullptr))
                return rc;
        }
    }
    return 0;
}


// ---- the parameter-gradient GEMMs of a step below 200,000 rows: one launch for all layers (inr_debug_set(20, 0): one per layer) --
// dW_l = dz_l^T act_l depends on the stashed activations and on dz_l only: all of them can wait until the input-gradient chain
// has produced every dz, and then run as ONE launch (gemm_hp_rc_multi_kernel) whose row splits are chosen for the launch as a
// whole -- one round of blocks over the chip -- instead of for every layer alone.  A rocprofv3 timeline of a 4,096-row step
// (tools/kt_timeline.py) had shown four 13-20 us launches that each fill half the chip, 28 us of event hand-shakes of the first
// form of this switch (a second stream), and 16 row splits per layer = 448 blocks writing 57 MB of slabs for `finalize` to read
// back.  Measured, ms per step, one launch per layer / merged (tools/side_stream_ab.py): 2,048 rows 0.182 / 0.146, 4,096: 0.216
// / 0.181, 8,192: 0.325 / 0.277, 16,384: 0.454 / 0.381, 32,768: 0.674 / 0.630, 69,632: 1.318 / 1.279, 139,264: 2.431 / 2.408,
// 262,144: 4.42 / 4.42.  (The last bits of the gradients differ between the two forms: other row ranges per partial sum.)
tune_int g_hp_side_stream{1};   // (the name of the first form; key 20) identical bits either way
tune_int g_hp_merge_blocks{256};  // key 22: block count the merged parameter-gradient launch aims at (row splits = this / tiles)

// the same step on the pre-split path (gemm_hp.inc): act[l] (l >= 1) and dz are HL32, act[0] = the HL32 image of x,
// dact fp32 until the backward pass overwrites it with dz (HL32, scaled from an a-priori bound).  Gradients are NOT reduced
// here: every producer leaves its slab rows in `slabs` and `fin` describes them (the caller finishes with launch_finalize).
// forward of the sine layers (stash kept for the backward pass); z_head: the last layer stashes z + b only (HPE_Z)
static int hp_forward_pass(const inr_siren_desc_t* d, const Layout& L, const float* params, std::vector<float*>& act,
                           std::vector<float*>& dact, const char* xhl, int64_t n, hipStream_t st, const H3Ctx& ctx, bool z_head,
                           bool skip_last = false) {
    // skip_last: the last sine layer runs inside hp_backward_pass with the head step in its epilogue (hp_sine_forward_head)
    const HpNet net{&ctx, &L};
    const int head = L.n_sine;
    auto act_hl = [&](int l) -> const char* { return l == 0 ? xhl : reinterpret_cast<const char*>(act[l]); };
    for (int l = 0; l < L.n_sine - (skip_last ? 1 : 0); ++l) {
        const float omega = (l == 0) ? d->first_omega : d->hidden_omega;
        if (int rc = hp_sine_forward(reinterpret_cast<char*>(act[l + 1]), dact[l], act_hl(l), net.w_hl(l), params + L.b_off[l], n,
                                     L.fan_in[l], L.fan_out[l], omega, net.act_scale(l), net.w_scale(l), 0, st,
                                     z_head && l == head - 1, net.act_scale(l + 1)))
            return rc;
    }
    return 0;
}

// head step + backward of every layer.  The loss gradient is formed here from (target, weight) -- the fused fit -- or taken from
// the caller (g_ext = dL/dy, the autograd path; target / weight unused, no loss).  Gradients are NOT reduced here: every producer
// leaves its slab rows in `slabs` and `fin` describes them (the caller finishes with launch_finalize).
static int hp_backward_pass(const inr_siren_desc_t* d, const Layout& L, const float* params, float* grads,
                            std::vector<float*>& act, std::vector<float*>& dact, const char* xhl, float* slabs,
                            const float* target, const float* weight, const float* g_ext, int64_t n, int64_t count_total,
                            float* loss_dst, hipStream_t st, const H3Ctx& ctx, FinalizeJob& fin, bool z_head, bool fuse_head = false) {
    const int H = d->hidden_features, head = L.n_sine;
    const HpNet net{&ctx, &L};
    const HpSlabPlan plan = hp_slab_plan(L, n);
    const float inv = (float)(1.0 / (double)(count_total > 0 ? count_total : n));
    const float omega_last = (head - 1 == 0) ? d->first_omega : d->hidden_omega;
    fin = FinalizeJob{};
    fin.nseg = 2 * (head + 1);
    for (int k = 0; k < fin.nseg; ++k) {
        const int l = k >> 1;
        fin.seg[k].slab = slabs + plan.seg[k].slab;
        fin.seg[k].stage1 = slabs + plan.seg[k].stage1;
        fin.seg[k].len = plan.seg[k].len;
        fin.seg[k].dst = (k & 1) ? L.b_off[l] : L.w_off[l];
        fin.seg[k].nslabs = 0;
    }
    fin.grads = grads;
    fin.loss_out = loss_dst;
    fin.loss_scale = inv;
    auto act_hl = [&](int l) -> const char* { return l == 0 ? xhl : reinterpret_cast<const char*>(act[l]); };
    // scale of dz_l: the head's bound for the last sine layer, else measured max|dz_{l+1}| * wnorm_{l+1} * omega_l
    auto dz_scale = [&](int l) {
        HpScale s;
        if (l == head - 1) {
            s.wn = net.head_bound();
            s.mul = 1.f;
        } else {
            s.meas = ctx.slots + 8 + l + 1;
            s.wn = reinterpret_cast<const float*>(ctx.slots + 16 + l + 1);
            s.mul = fabsf(l == 0 ? d->first_omega : d->hidden_omega) * 1.001f;
        }
        return s;
    };
    if (fuse_head) {   // the last sine layer with the head step in its epilogue: no z round trip, no head step kernel (gemm_hp_row.inc, HPE_HEAD)
        const int rows = hp_row_head_rows(n);
        const int kb = 2 * (head - 1) + 1, kw = 2 * head, kg = 2 * head + 1;   // b_{S-1}, W_head, b_head
        float* part_loss = slabs + plan.part_loss;
        if (int rc = hp_sine_forward_head(reinterpret_cast<char*>(dact[head - 1]), act_hl(head - 1), net.w_hl(head - 1),
                                          params + L.b_off[head - 1], n, L.fan_in[head - 1], H, omega_last, net.act_scale(head - 1),
                                          net.w_scale(head - 1), dz_scale(head - 1), params + L.w_off[head], params + L.b_off[head],
                                          target, weight, count_total, const_cast<float*>(fin.seg[kb].slab),
                                          const_cast<float*>(fin.seg[kw].slab), part_loss, const_cast<float*>(fin.seg[kg].slab),
                                          ctx.slots + 8 + head - 1, st))
            return rc;
        fin.seg[kb].nslabs = fin.seg[kw].nslabs = fin.seg[kg].nslabs = rows;
        fin.part_loss = part_loss;
        fin.nparts = rows;
    } else {
        const int blocks = (int)hp_head_blocks(n);
        const int kb = 2 * (head - 1) + 1, kw = 2 * head, kg = 2 * head + 1;   // b_{S-1}, W_head, b_head
        float* part_loss = slabs + plan.part_loss;
        if (int rc = hp_head_step(reinterpret_cast<char*>(dact[head - 1]), const_cast<float*>(fin.seg[kb].slab),
                                  const_cast<float*>(fin.seg[kw].slab), part_loss, const_cast<float*>(fin.seg[kg].slab),
                                  act_hl(head), dact[head - 1], params + L.w_off[head], params + L.b_off[head], target, weight,
                                  n, H, count_total, ctx.slots + 8 + head - 1, dz_scale(head - 1), st, z_head, omega_last, g_ext,
                                  net.act_scale(head)))
            return rc;
        fin.seg[kb].nslabs = fin.seg[kw].nslabs = fin.seg[kg].nslabs = blocks;
        fin.part_loss = part_loss;
        fin.nparts = blocks;
    }
    // (only where it pays: see above)
    const bool merge = g_hp_side_stream && L.n_sine > 1 && n < 200000 && L.n_sine <= hp_param_grad_multi_max();
    HpParamGradJob jobs[16];
    int njobs = 0;
    // merged: the launch as a whole should be ONE round of blocks over the chip -- row splits for that, not for each layer alone
    // (at 4,096 rows 16 splits per layer made 448 blocks that wrote 57 MB of slabs for `finalize` to read back; 9 make 252 and 32)
    int merged_splits = 1 << 30;
    if (merge) {
        long long tiles_all = 0;
        for (int l = 0; l < L.n_sine; ++l) tiles_all += (long long)((L.fan_out[l] + 127) / 128) * ((L.fan_in[l] + 255) / 256);
        const int target = g_hp_merge_blocks;
        merged_splits = (int)(target / tiles_all > 1 ? target / tiles_all : 1);
        const int min_by_len = (int)((n + 16383) / 16384);      // (hp_param_grad_splits: no register accumulation over > 16 k rows)
        if (merged_splits < min_by_len) merged_splits = min_by_len;
    }
    for (int l = L.n_sine - 1; l >= 0; --l) {
        const char* dz = reinterpret_cast<const char*>(dact[l]);
        int splits = hp_param_grad_splits(n, L.fan_in[l], L.fan_out[l]);     // (what the slab plan reserves: an upper bound)
        if (splits > merged_splits) splits = merged_splits;
        if (merge) {
            jobs[njobs++] = HpParamGradJob{const_cast<float*>(fin.seg[2 * l].slab), splits, dz, act_hl(l), L.fan_in[l], L.fan_out[l],
                                           dz_scale(l), net.act_scale(l)};
        } else if (int rc = hp_param_grad_slabs(const_cast<float*>(fin.seg[2 * l].slab), splits, dz, act_hl(l), n, L.fan_in[l],
                                                L.fan_out[l], dz_scale(l), net.act_scale(l), st)) {
            return rc;
        }
        fin.seg[2 * l].nslabs = splits;
        if (l > 0) {
            int rows = 0;
            if (int rc = hp_input_grad(reinterpret_cast<char*>(dact[l - 1]), dz, net.wT_hl(l), dact[l - 1], n, L.fan_in[l],
                                       L.fan_out[l], const_cast<float*>(fin.seg[2 * (l - 1) + 1].slab), &rows,
                                       ctx.slots + 8 + l - 1, dz_scale(l), net.w_scale(l), dz_scale(l - 1), st))
                return rc;
            fin.seg[2 * (l - 1) + 1].nslabs = rows;
        }
    }
    if (merge) {
        if (int rc = hp_param_grad_multi(jobs, njobs, n, st)) return rc;
    }
    return 0;
}

static int fit_forward_backward_hp(const inr_siren_desc_t* d, const Layout& L, const float* params, float* grads,
                                   std::vector<float*>& act, std::vector<float*>& dact, const char* xhl, float* slabs,
                                   const float* target, const float* weight, int64_t n, int64_t count_total, float* loss_dst,
                                   hipStream_t st, const H3Ctx& ctx, FinalizeJob& fin) {
    const int head = L.n_sine;
    const HpNet net{&ctx, &L};
    const float inv = (float)(1.0 / (double)(count_total > 0 ? count_total : n));
    const float omega_last = (head - 1 == 0) ? d->first_omega : d->hidden_omega;
    const HpHeadBoundArgs hb{params + L.w_off[head], params + L.b_off[head], ctx.slots + 25, weight ? ctx.slots + 26 : nullptr,
                             inv, omega_last};
    if (int rc = hp_refresh_weights(net, d, params, st, true, &hb)) return rc;
    // the output of the LAST sine layer feeds nothing but the head: that layer stashes z + b only (one fp32 matrix instead
    // of act + omega cos) and the head step forms sin / cos itself -- 2.1 GB less HBM traffic per step at N = 524,288
    const bool z_head = hp_z_stash_ok(L.fan_in[head - 1]);
    // ... and from 768 row panels (98,304 rows) on the head step rides in that layer's epilogue (a block of gemm_hp_row_kernel owns whole rows)
    const bool fuse_head = z_head && hp_row_head_ok(n, d->hidden_features, L.fan_in[head - 1]);
    if (int rc = hp_forward_pass(d, L, params, act, dact, xhl, n, st, ctx, z_head, fuse_head)) return rc;
    return hp_backward_pass(d, L, params, grads, act, dact, xhl, slabs, target, weight, nullptr, n, count_total, loss_dst, st, ctx,
                            fin, z_head, fuse_head);
}

// per call: scales of the network input and of the targets, HL32 image of x
static int hp_prepare_call(const H3Ctx& ctx, const Layout& L, char* xhl, const float* x, const float* target,
                           const float* weight, int64_t n, int out_f, hipStream_t st, int64_t n_acq = 1, bool keep_x = false,
                           bool keep_stats = false) {
    // (several acquisitions behind one pointer: the bounds cover all of them)
    if (!keep_stats) {
        if (int rc = h3_tensor_amax(ctx.slots + 25, target, (long long)n * n_acq * out_f, st, 0u)) return rc;
        if (weight) {
            if (int rc = h3_tensor_amax(ctx.slots + 26, weight, (long long)n * n_acq * out_f, st, 0u)) return rc;
        }
    }
    if (keep_x) return 0;     // the HL32 image of x and its scale slot are the previous call's (the caller vouches for it)
    HpScale sx;
    sx.meas = ctx.slots + 24;
    sx.mul = 1.f;
    return hp_convert(xhl, x, n, L.fan_in[0], sx, st);
}

// What the last inr_siren_loss_grad_ex call left in a workspace, remembered on the HOST (the call must not sync to look into the
// device bytes): a REUSE flag is honoured only when this call's (n, fan_in, x) -- and (target, weight) for the statistics -- are
// the ones the image / slots in that workspace were built from.  A wrong flag (first call, another n, another x) used to run on
// stale or uninitialised operand images and return wrong gradients with rc 0.
namespace {
struct ReuseStamp {
    const void* ws = nullptr;
    int64_t n = 0;
    int fan_in = 0;
    const void *x = nullptr, *target = nullptr, *weight = nullptr;
    bool image = false, stats = false;
    bool fwd_train = false;      // an inr_siren_forward_train stash is pending (inr_siren_backward_train consumes it)
    unsigned long long used = 0; // last use (least recently used entry is replaced)
};
std::mutex g_reuse_mu;
ReuseStamp g_reuse[64];
unsigned long long g_reuse_next = 0;
ReuseStamp* reuse_find(const void* ws, bool create) {      // (caller holds g_reuse_mu)
    for (auto& s : g_reuse)
        if (s.ws == ws) {
            s.used = ++g_reuse_next;
            return &s;
        }
    if (!create) return nullptr;
    ReuseStamp* lru = &g_reuse[0];                          // the least recently used entry makes room
    for (auto& s : g_reuse)
        if (s.used < lru->used) lru = &s;
    *lru = ReuseStamp{};
    lru->ws = ws;
    lru->used = ++g_reuse_next;
    return lru;
}
}   // namespace

size_t inr_siren_fit_workspace_bytes(const inr_siren_desc_t* desc, int64_t n) {
    if (check_desc(desc) || n < 1) return 0;
    const Layout L = make_layout(desc);
    return fit_carve(desc, L, n).total;
}

int inr_siren_fit(const inr_siren_desc_t* desc, float* params, float* grads, float* m, float* v, const float* x,
                  const float* target, const float* weight, int64_t n, int64_t first_step, int n_steps, double lr,
                  double beta1, double beta2, double eps, float* losses, void* workspace, size_t workspace_bytes,
                  void* stream) {
    return inr_siren_fit_cycle(desc, params, grads, m, v, x, target, weight, 1, 0, n, first_step, n_steps, lr, beta1, beta2,
                               eps, losses, workspace, workspace_bytes, stream);
}

int inr_siren_fit_cycle(const inr_siren_desc_t* desc, float* params, float* grads, float* m, float* v, const float* x,
                        const float* targets, const float* weights, int n_acq, int first_acq, int64_t n,
                        int64_t first_step, int n_steps, double lr, double beta1, double beta2, double eps, float* losses,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(desc)) return rc;
    const float* target = targets;
    const float* weight = weights;
    INR_REQUIRE(params && grads && m && v && x && target, INR_E_INVALID, "inr_siren_fit: null pointer");
    INR_REQUIRE(n >= 1 && n <= MAX_ROWS, INR_E_INVALID, "inr_siren_fit: bad row count %lld", (long long)n);
    INR_REQUIRE(first_step >= 1 && n_steps >= 0, INR_E_INVALID, "inr_siren_fit: first_step >= 1, n_steps >= 0");
    INR_REQUIRE(n_acq >= 1 && first_acq >= 0 && first_acq < n_acq, INR_E_INVALID,
                "inr_siren_fit_cycle: need n_acq >= 1 and 0 <= first_acq < n_acq");
    const int64_t acq_stride = n * desc->out_features;      // floats between consecutive acquisitions
    const Layout L = make_layout(desc);
    const FitCarve c = fit_carve(desc, L, n);
    INR_REQUIRE(workspace && workspace_bytes >= c.total, INR_E_WORKSPACE,
                "inr_siren_fit: workspace too small (%zu < %zu)", workspace_bytes, c.total);
    INR_REQUIRE(aligned16(workspace) && aligned16(params) && aligned16(grads) && aligned16(x), INR_E_ALIGN,
                "inr_siren_fit: params/grads/x/workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace;
    {   // this call rebuilds the operand image and the statistics slots of the workspace: a later REUSE flag must not trust them
        std::lock_guard<std::mutex> lk(g_reuse_mu);
        if (ReuseStamp* s = reuse_find(workspace, false)) s->image = s->stats = s->fwd_train = false;
    }
    std::vector<float*> act(L.n_sine + 1), dact(L.n_sine);
    act[0] = const_cast<float*>(x);
    for (int l = 0; l < L.n_sine; ++l) {
        act[l + 1] = (float*)(base + (size_t)l * c.act_b);
        dact[l] = (float*)(base + (size_t)(L.n_sine + l) * c.act_b);
    }
    float* y = (float*)(base + 2 * (size_t)L.n_sine * c.act_b);
    float* gy = (float*)((char*)y + c.out_b);
    float* scratch = (float*)((char*)gy + c.out_b);
    float* loss_sink = scratch;  // overwritten later in the step; only used when losses == nullptr
    if (small_path_ok(desc, n) && !g_force_generic) {
        // master.py regime (small network, few thousand rows): one fused forward+backward launch + one reduce/Adam
        // launch per step instead of ~45 layer-wise launches (csrc/siren_small.hip)
        long long w_off[32], b_off[32];
        for (int l = 0; l <= L.n_sine; ++l) { w_off[l] = L.w_off[l]; b_off[l] = L.b_off[l]; }
        if (g_small_multi && small_multi_ok(desc, n)) {   // all steps inside one persistent launch per 64 steps
            const int rc = small_fit_multi(desc, w_off, b_off, L.total, params, grads, m, v, x, targets, weights, n_acq,
                                           first_acq, n, first_step, n_steps, lr, beta1, beta2, eps, losses,
                                           (float*)workspace, st);
            if (rc != INR_E_FALLBACK) return rc;   // (a device that cannot hold the grid co-resident: two launches per step)
        }
        for (int it = 0; it < n_steps; ++it) {
            const int64_t a = (first_acq + it) % n_acq;
            if (int rc = small_fit_step(desc, w_off, b_off, L.total, params, grads, m, v, x, targets + a * acq_stride,
                                        weights ? weights + a * acq_stride : nullptr, n, first_step + it, lr, beta1,
                                        beta2, eps, losses ? losses + it : nullptr, (float*)workspace, st))
                return rc;
        }
        return 0;
    }

    H3Ctx h3;
    if (h3_eligible(L)) {
        h3 = h3_make_ctx(L, base + c.h3_off);
        if (int rc = h3_tensor_amax(h3.slots + 24, x, (long long)n * L.fan_in[0], st, 0x3f800000u)) return rc;
    }
    const bool hp = hp_eligible(desc, L);
    if (hp) loss_sink = scratch + hp_slab_plan(L, n).loss_sink;   // (scratch[0] is a slab row the finalize launch still reads)
    char* xhl = base + c.xhl_off;
    if (hp && n_steps > 0) {
        if (int rc = hp_prepare_call(h3, L, xhl, x, targets, weights, n, desc->out_features, st, n_acq)) return rc;
    }
    for (int it = 0; it < n_steps; ++it) {
        const int64_t a = (first_acq + it) % n_acq;
        target = targets + a * acq_stride;
        weight = weights ? weights + a * acq_stride : nullptr;
        if (hp) {   // gradient sums, loss and the Adam step in one launch behind the backward pass
            FinalizeJob fin;
            if (int rc = fit_forward_backward_hp(desc, L, params, grads, act, dact, xhl, scratch, target, weight, n, 0,
                                                 losses ? (losses + it) : loss_sink, st, h3, fin))
                return rc;
            fin.params = params;
            fin.m = m;
            fin.v = v;
            if (int rc = launch_finalize(fin, first_step + it, lr, beta1, beta2, eps, st)) return rc;
            continue;
        }
        if (int rc = fit_forward_backward(desc, L, params, grads, act, dact, y, gy, scratch, target, weight, n, 0,
                                          losses ? (losses + it) : loss_sink, st, &h3))
            return rc;
        if (int rc = launch_adam(params, grads, m, v, L.total, first_step + it, lr, beta1, beta2, eps, st)) return rc;
    }
    return 0;
}

// forward + loss + backward only (no optimizer): the building block of a fit whose rows are split over GPUs
int inr_siren_loss_grad(const inr_siren_desc_t* desc, const float* params, float* grads, const float* x,
                        const float* target, const float* weight, int64_t n, int64_t count_total, float* loss,
                        void* workspace, size_t workspace_bytes, void* stream) {
    return inr_siren_loss_grad_ex(desc, params, grads, x, target, weight, n, count_total, loss, workspace, workspace_bytes, 0,
                                  stream);
}


int inr_siren_loss_grad_ex(const inr_siren_desc_t* desc, const float* params, float* grads, const float* x,
                           const float* target, const float* weight, int64_t n, int64_t count_total, float* loss,
                           void* workspace, size_t workspace_bytes, int flags, void* stream) {
    if (int rc = check_desc(desc)) return rc;
    INR_REQUIRE((flags & ~(INR_REUSE_INPUT_IMAGE | INR_REUSE_TARGET_STATS)) == 0, INR_E_INVALID,
                "inr_siren_loss_grad_ex: unknown flags 0x%x", flags);
    INR_REQUIRE(params && grads && x && target && loss, INR_E_INVALID, "inr_siren_loss_grad: null pointer");
    INR_REQUIRE(n >= 1 && n <= MAX_ROWS, INR_E_INVALID, "inr_siren_loss_grad: bad row count %lld", (long long)n);
    INR_REQUIRE(count_total == 0 || count_total >= n * desc->out_features, INR_E_INVALID,
                "inr_siren_loss_grad: count_total must be 0 or >= n*out_features");
    const Layout L = make_layout(desc);
    const FitCarve c = fit_carve(desc, L, n);
    INR_REQUIRE(workspace && workspace_bytes >= c.total, INR_E_WORKSPACE,
                "inr_siren_loss_grad: workspace too small (%zu < %zu)", workspace_bytes, c.total);
    INR_REQUIRE(aligned16(workspace) && aligned16(params) && aligned16(grads) && aligned16(x), INR_E_ALIGN,
                "inr_siren_loss_grad: params/grads/x/workspace must be 16-byte aligned");
    char* base = (char*)workspace;
    std::vector<float*> act(L.n_sine + 1), dact(L.n_sine);
    act[0] = const_cast<float*>(x);
    for (int l = 0; l < L.n_sine; ++l) {
        act[l + 1] = (float*)(base + (size_t)l * c.act_b);
        dact[l] = (float*)(base + (size_t)(L.n_sine + l) * c.act_b);
    }
    float* y = (float*)(base + 2 * (size_t)L.n_sine * c.act_b);
    float* gy = (float*)((char*)y + c.out_b);
    float* scratch = (float*)((char*)gy + c.out_b);
    H3Ctx h3;
    // A network the pre-split kernels do not serve keeps no operand image and no target statistics in its workspace: there is
    // nothing a REUSE flag could refer to, so the flags are accepted and mean nothing (round 5: they were REFUSED, which made every
    // multi-step ShardedSirenFitter.step on e.g. Siren(32, 64, 1, 1) fail at its second step -- found by tests/test_gpu_nccl.py).
    if (!hp_eligible(desc, L)) flags = 0;
    const bool keep_x = (flags & INR_REUSE_INPUT_IMAGE) != 0;
    {
        std::lock_guard<std::mutex> lk(g_reuse_mu);
        ReuseStamp* s = reuse_find(workspace, flags == 0);
        if (flags & INR_REUSE_INPUT_IMAGE)
            INR_REQUIRE(s && s->image && s->n == n && s->fan_in == L.fan_in[0] && s->x == x, INR_E_INVALID,
                        "inr_siren_loss_grad_ex: INR_REUSE_INPUT_IMAGE, but this workspace does not hold the image of these %lld rows of x",
                        (long long)n);
        if (flags & INR_REUSE_TARGET_STATS)
            INR_REQUIRE(s && s->stats && s->n == n && s->target == target && s->weight == weight, INR_E_INVALID,
                        "inr_siren_loss_grad_ex: INR_REUSE_TARGET_STATS, but this workspace does not hold the statistics of this target");
        if (!s) s = reuse_find(workspace, true);
        const bool hp_path = hp_eligible(desc, L);
        if (!(flags & INR_REUSE_INPUT_IMAGE)) { s->image = hp_path; s->x = x; s->fan_in = L.fan_in[0]; }
        if (!(flags & INR_REUSE_TARGET_STATS)) { s->stats = hp_path; s->target = target; s->weight = weight; }
        s->n = n;
    }
    if (h3_eligible(L)) {
        h3 = h3_make_ctx(L, base + c.h3_off);
        if (!keep_x) {
            if (int rc = h3_tensor_amax(h3.slots + 24, x, (long long)n * L.fan_in[0], (hipStream_t)stream, 0x3f800000u)) return rc;
        }
    }
    if (hp_eligible(desc, L)) {
        char* xhl = base + c.xhl_off;
        if (int rc = hp_prepare_call(h3, L, xhl, x, target, weight, n, desc->out_features, (hipStream_t)stream, 1, keep_x,
                                     (flags & INR_REUSE_TARGET_STATS) != 0))
            return rc;
        FinalizeJob fin;
        if (int rc = fit_forward_backward_hp(desc, L, params, grads, act, dact, xhl, scratch, target, weight, n, count_total, loss,
                                             (hipStream_t)stream, h3, fin))
            return rc;
        return launch_finalize(fin, 0, 0.0, 0.0, 0.0, 0.0, (hipStream_t)stream);
    }
    return fit_forward_backward(desc, L, params, grads, act, dact, y, gy, scratch, target, weight, n, count_total, loss,
                                (hipStream_t)stream, &h3);
}

// ---- the autograd path on the fused fit's kernels (include/inrhip.h (f)) ------------------------------------------------------
int inr_siren_hp_eligible(const inr_siren_desc_t* desc) {
    if (check_desc(desc)) return 0;
    const Layout L = make_layout(desc);
    return hp_eligible(desc, L) ? 1 : 0;
}

namespace {
struct TrainCarve {
    std::vector<float*> act, dact;
    float* scratch;
    char* xhl;
    H3Ctx h3;
};
// (the carve of inr_siren_fit: both passes of a step must see the same one)
int train_carve(TrainCarve& t, const inr_siren_desc_t* desc, const Layout& L, const float* x, int64_t n, void* workspace,
                size_t workspace_bytes, const char* who) {
    const FitCarve c = fit_carve(desc, L, n);
    INR_REQUIRE(workspace && workspace_bytes >= c.total, INR_E_WORKSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes,
                c.total);
    char* base = (char*)workspace;
    t.act.assign(L.n_sine + 1, nullptr);
    t.dact.assign(L.n_sine, nullptr);
    t.act[0] = const_cast<float*>(x);
    for (int l = 0; l < L.n_sine; ++l) {
        t.act[l + 1] = (float*)(base + (size_t)l * c.act_b);
        t.dact[l] = (float*)(base + (size_t)(L.n_sine + l) * c.act_b);
    }
    float* y = (float*)(base + 2 * (size_t)L.n_sine * c.act_b);
    float* gy = (float*)((char*)y + c.out_b);
    t.scratch = (float*)((char*)gy + c.out_b);
    t.xhl = base + c.xhl_off;
    t.h3 = h3_make_ctx(L, base + c.h3_off);
    return 0;
}
}   // namespace

int inr_siren_forward_train(const inr_siren_desc_t* desc, const float* params, const float* x, float* y, int64_t n,
                            void* workspace, size_t workspace_bytes, int flags, void* stream) {
    if (int rc = check_desc(desc)) return rc;
    INR_REQUIRE((flags & ~INR_REUSE_INPUT_IMAGE) == 0, INR_E_INVALID, "inr_siren_forward_train: unknown flags 0x%x", flags);
    INR_REQUIRE(params && x && y, INR_E_INVALID, "inr_siren_forward_train: null pointer");
    INR_REQUIRE(n >= 1 && n <= MAX_ROWS, INR_E_INVALID, "inr_siren_forward_train: bad row count %lld", (long long)n);
    const Layout L = make_layout(desc);
    INR_REQUIRE(hp_eligible(desc, L), INR_E_INVALID, "inr_siren_forward_train: network shape not served by the pre-split kernels "
                "(ask inr_siren_hp_eligible)");
    INR_REQUIRE(aligned16(workspace) && aligned16(params) && aligned16(x), INR_E_ALIGN,
                "inr_siren_forward_train: params/x/workspace must be 16-byte aligned");
    TrainCarve t;
    if (int rc = train_carve(t, desc, L, x, n, workspace, workspace_bytes, "inr_siren_forward_train")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool keep_x = (flags & INR_REUSE_INPUT_IMAGE) != 0;
    {
        std::lock_guard<std::mutex> lk(g_reuse_mu);
        ReuseStamp* s = reuse_find(workspace, !keep_x);
        if (keep_x)
            INR_REQUIRE(s && s->image && s->n == n && s->fan_in == L.fan_in[0] && s->x == x, INR_E_INVALID,
                        "inr_siren_forward_train: INR_REUSE_INPUT_IMAGE, but this workspace does not hold the image of these %lld rows of x",
                        (long long)n);
        s->image = true;
        s->x = x;
        s->fan_in = L.fan_in[0];
        s->n = n;
        s->stats = false;
        s->fwd_train = true;
    }
    if (!keep_x) {
        if (int rc = h3_tensor_amax(t.h3.slots + 24, x, (long long)n * L.fan_in[0], st, 0x3f800000u)) return rc;
        HpScale sx;
        sx.meas = t.h3.slots + 24;
        sx.mul = 1.f;
        if (int rc = hp_convert(t.xhl, x, n, L.fan_in[0], sx, st)) return rc;
    }
    const HpNet net{&t.h3, &L};
    if (int rc = hp_refresh_weights(net, desc, params, st, true, nullptr)) return rc;      // (also zeroes this step's dz maxima)
    const int head = L.n_sine;
    const bool z_head = hp_z_stash_ok(L.fan_in[head - 1]);
    if (int rc = hp_forward_pass(desc, L, params, t.act, t.dact, t.xhl, n, st, t.h3, z_head)) return rc;
    const float omega_last = (head - 1 == 0) ? desc->first_omega : desc->hidden_omega;
    return hp_head_forward(y, z_head ? reinterpret_cast<const char*>(t.dact[head - 1]) : reinterpret_cast<const char*>(t.act[head]),
                           params + L.w_off[head], params + L.b_off[head], n, desc->hidden_features, 0, 0.f, st, z_head, omega_last,
                           net.act_scale(head));
}

int inr_siren_backward_train(const inr_siren_desc_t* desc, const float* params, float* grads, const float* gy, int64_t n,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(desc)) return rc;
    INR_REQUIRE(params && grads && gy, INR_E_INVALID, "inr_siren_backward_train: null pointer");
    INR_REQUIRE(n >= 1 && n <= MAX_ROWS, INR_E_INVALID, "inr_siren_backward_train: bad row count %lld", (long long)n);
    const Layout L = make_layout(desc);
    INR_REQUIRE(hp_eligible(desc, L), INR_E_INVALID, "inr_siren_backward_train: network shape not served by the pre-split kernels");
    INR_REQUIRE(aligned16(workspace) && aligned16(params) && aligned16(grads), INR_E_ALIGN,
                "inr_siren_backward_train: params/grads/workspace must be 16-byte aligned");
    const void* x = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reuse_mu);
        ReuseStamp* s = reuse_find(workspace, false);
        INR_REQUIRE(s && s->fwd_train && s->n == n, INR_E_INVALID,
                    "inr_siren_backward_train: no inr_siren_forward_train of %lld rows is pending on this workspace", (long long)n);
        s->fwd_train = false;       // (the backward overwrites the stash with dz: one backward per forward)
        x = s->x;
    }
    TrainCarve t;
    if (int rc = train_carve(t, desc, L, (const float*)x, n, workspace, workspace_bytes, "inr_siren_backward_train")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int head = L.n_sine;
    const bool z_head = hp_z_stash_ok(L.fan_in[head - 1]);
    const HpNet net{&t.h3, &L};
    const float omega_last = (head - 1 == 0) ? desc->first_omega : desc->hidden_omega;
    // the scale of the head's dz: max|gy| x max|w_head| x omega (measured: gy is the caller's)
    if (int rc = h3_tensor_amax(t.h3.slots + 25, gy, (long long)n * desc->out_features, st, 0u)) return rc;
    if (int rc = hp_head_bound_ext(net.head_bound(), t.h3.slots + 25, params + L.w_off[head], desc->hidden_features, omega_last, st))
        return rc;
    FinalizeJob fin;
    float* loss_sink = t.scratch + hp_slab_plan(L, n).loss_sink;
    if (int rc = hp_backward_pass(desc, L, params, grads, t.act, t.dact, t.xhl, t.scratch, nullptr, nullptr, gy, n, 1, loss_sink, st,
                                  t.h3, fin, z_head))
        return rc;
    return launch_finalize(fin, 0, 0.0, 0.0, 0.0, 0.0, st);
}

// ---- metrics ---------------------------------------------------------------------------------------------
size_t inr_metric_workspace_bytes(int n_images) {
    return (size_t)metric_workspace_doubles(n_images > 0 ? n_images : 1) * sizeof(double);
}

int inr_psnr(double* out, const float* x, const float* y, int n_images, int64_t per_image, double data_range,
             void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(out && x && y, INR_E_INVALID, "inr_psnr: null pointer");
    INR_REQUIRE(n_images >= 1 && n_images <= 65535 && per_image >= 1 && data_range > 0, INR_E_INVALID,
                "inr_psnr: bad sizes");
    INR_REQUIRE(workspace && workspace_bytes >= inr_metric_workspace_bytes(n_images), INR_E_WORKSPACE,
                "inr_psnr: workspace too small");
    return launch_psnr(out, x, y, n_images, per_image, data_range, (double*)workspace, (hipStream_t)stream);
}

int inr_ssim2d(double* out, const float* x, const float* y, int n_images, int height, int width, int win,
               double data_range, int use_mask, float mask_thr, void* workspace, size_t workspace_bytes,
               void* stream) {
    INR_REQUIRE(out && x && y, INR_E_INVALID, "inr_ssim2d: null pointer");
    INR_REQUIRE(n_images >= 1 && n_images <= 65535 && win >= 3 && (win & 1) && height >= win && width >= win &&
                    data_range > 0,
                INR_E_INVALID, "inr_ssim2d: need odd win >= 3 and images at least win x win (got %dx%d, win %d)",
                height, width, win);
    INR_REQUIRE(workspace && workspace_bytes >= inr_metric_workspace_bytes(n_images), INR_E_WORKSPACE,
                "inr_ssim2d: workspace too small");
    return launch_ssim(out, x, y, n_images, height, width, win, data_range, use_mask, mask_thr, (double*)workspace,
                       (hipStream_t)stream);
}

int inr_hybrid_fit(double* params, int* status, int* nfev, double* cost, const double* signals, int64_t n_voxels,
                   void* stream) {
    if (n_voxels == 0) return 0;
    INR_REQUIRE(params && status && nfev && cost && signals, INR_E_INVALID, "inr_hybrid_fit: null pointer");
    // (one 8-lane group per voxel, grid dimension is 32-bit: 2^31 - 1 blocks of 8 voxels)
    INR_REQUIRE(n_voxels > 0 && n_voxels <= ((int64_t)1 << 33), INR_E_INVALID, "inr_hybrid_fit: bad voxel count %lld",
                (long long)n_voxels);
    return launch_hybrid_fit(params, status, nfev, cost, signals, n_voxels, (hipStream_t)stream);
}

size_t inr_rams_shift_loss_workspace_bytes(int n_images, int border) {
    const int ns = 2 * (border > 0 ? border : 0) + 1;
    return (size_t)(n_images > 0 ? n_images : 1) * ns * ns * sizeof(double);
}

int inr_rams_shift_loss(double* out, const float* y_true, const float* y_pred, const float* mask, int n_images, int size,
                        int border, int mode, void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(out && y_true && y_pred && mask, INR_E_INVALID, "inr_rams_shift_loss: null pointer");
    INR_REQUIRE(n_images >= 1 && n_images <= 65535 && border >= 0 && border <= 16 && size > 2 * border &&
                    (mode == 0 || mode == 1),
                INR_E_INVALID, "inr_rams_shift_loss: bad arguments (size=%d border=%d mode=%d)", size, border, mode);
    INR_REQUIRE(workspace && workspace_bytes >= inr_rams_shift_loss_workspace_bytes(n_images, border), INR_E_WORKSPACE,
                "inr_rams_shift_loss: workspace too small");
    return launch_shift_loss(out, y_true, y_pred, mask, n_images, size, border, mode, (double*)workspace,
                             (hipStream_t)stream);
}

static int conv3d_dims_ok(int B, int D1, int D2, int D3, int pad) {
    INR_REQUIRE(B >= 1 && B <= 65535 && D1 >= 1 && D2 >= 1 && D3 >= 1 && (pad == 0 || pad == 1), INR_E_INVALID,
                "conv3d: bad shape (B=%d D=%dx%dx%d pad=%d)", B, D1, D2, D3, pad);
    INR_REQUIRE(D1 + 2 * pad > 2 && D2 + 2 * pad > 2 && D3 + 2 * pad > 2, INR_E_INVALID, "conv3d: volume smaller than the kernel");
    INR_REQUIRE((long long)B * D1 * D2 * D3 * 32 * 4 < (1ll << 40), INR_E_INVALID, "conv3d: volume too large");
    // the kernels address one image through a 32-bit buffer window
    INR_REQUIRE((long long)D1 * D2 * D3 * 32 * 4 < (1ll << 31), INR_E_INVALID, "conv3d: one image must stay below 2 GiB");
    return 0;
}

int inr_rams_conv3d_forward(float* y, const float* x, const float* w, const float* bias, int B, int D1, int D2, int D3, int pad,
                            int relu, void* stream) {
    INR_REQUIRE(y && x && w && bias, INR_E_INVALID, "inr_rams_conv3d_forward: null pointer");
    if (int rc = conv3d_dims_ok(B, D1, D2, D3, pad)) return rc;
    INR_REQUIRE(aligned16(x) && aligned16(w), INR_E_ALIGN, "inr_rams_conv3d_forward: x / w must be 16-byte aligned");
    return rams_conv3d_forward(y, x, w, bias, B, D1, D2, D3, pad, relu, (hipStream_t)stream);
}

size_t inr_rams_conv3d_dgrad_workspace_bytes(void) { return (27 * 32 * 32 + 64) * sizeof(float); }

int inr_rams_conv3d_dgrad(float* dx, const float* dy, const float* w, int B, int D1, int D2, int D3, void* workspace,
                          size_t workspace_bytes, void* stream) {
    INR_REQUIRE(dx && dy && w, INR_E_INVALID, "inr_rams_conv3d_dgrad: null pointer");
    if (int rc = conv3d_dims_ok(B, D1, D2, D3, 1)) return rc;
    INR_REQUIRE(workspace && workspace_bytes >= inr_rams_conv3d_dgrad_workspace_bytes(), INR_E_WORKSPACE,
                "inr_rams_conv3d_dgrad: workspace too small");
    INR_REQUIRE(aligned16(dy) && aligned16(workspace), INR_E_ALIGN, "inr_rams_conv3d_dgrad: dy / workspace must be 16-byte aligned");
    return rams_conv3d_dgrad_same(dx, dy, w, B, D1, D2, D3, (float*)workspace, (hipStream_t)stream);
}

size_t inr_rams_conv3d_wgrad_workspace_bytes(int B, int D1, int D2, int D3, int pad) {
    const long long nvox = (long long)B * (D1 + 2 * pad - 2) * (D2 + 2 * pad - 2) * (D3 + 2 * pad - 2);
    return rams_conv3d_wgrad_ws_floats(nvox > 0 ? nvox : 1) * sizeof(float);
}

int inr_rams_conv3d_wgrad(float* gw, float* gb, const float* x, const float* dy, int B, int D1, int D2, int D3, int pad,
                          void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(gw && x && dy, INR_E_INVALID, "inr_rams_conv3d_wgrad: null pointer");
    if (int rc = conv3d_dims_ok(B, D1, D2, D3, pad)) return rc;
    INR_REQUIRE(workspace && workspace_bytes >= inr_rams_conv3d_wgrad_workspace_bytes(B, D1, D2, D3, pad), INR_E_WORKSPACE,
                "inr_rams_conv3d_wgrad: workspace too small");
    INR_REQUIRE(aligned16(x) && aligned16(dy), INR_E_ALIGN, "inr_rams_conv3d_wgrad: x / dy must be 16-byte aligned");
    return rams_conv3d_wgrad_auto(gw, gb, x, dy, B, D1, D2, D3, pad, (float*)workspace, (hipStream_t)stream);
}

size_t inr_rams_shift_loss_grad_workspace_bytes(int n_images, int border) {
    return inr_rams_shift_loss_workspace_bytes(n_images, border) + (size_t)(n_images > 0 ? n_images : 1) * sizeof(int) + 8;
}

int inr_rams_shift_loss_grad(double* loss, float* grad_pred, const float* y_true, const float* y_pred, const float* mask,
                             const float* upstream, int n_images, int size, int border, void* workspace,
                             size_t workspace_bytes, void* stream) {
    INR_REQUIRE(loss && grad_pred && y_true && y_pred && mask, INR_E_INVALID, "inr_rams_shift_loss_grad: null pointer");
    INR_REQUIRE(n_images >= 1 && n_images <= 65535 && border >= 0 && border <= 16 && size > 2 * border, INR_E_INVALID,
                "inr_rams_shift_loss_grad: bad arguments (size=%d border=%d)", size, border);
    INR_REQUIRE(workspace && workspace_bytes >= inr_rams_shift_loss_grad_workspace_bytes(n_images, border), INR_E_WORKSPACE,
                "inr_rams_shift_loss_grad: workspace too small");
    INR_REQUIRE(((uintptr_t)workspace & 7) == 0, INR_E_ALIGN, "inr_rams_shift_loss_grad: workspace must be 8-byte aligned");
    return launch_shift_loss_grad(loss, grad_pred, y_true, y_pred, mask, upstream, n_images, size, border, (double*)workspace,
                                  (hipStream_t)stream);
}

int inr_adc_map(float* out, const float* data, const float* bvals, int64_t n_pixels, int n_b, void* stream) {
    INR_REQUIRE(out && data && bvals, INR_E_INVALID, "inr_adc_map: null pointer");
    INR_REQUIRE(n_pixels >= 0 && n_b >= 2 && n_b <= 32, INR_E_INVALID, "inr_adc_map: need 2 <= n_b <= 32");
    return launch_adc(out, data, bvals, n_pixels, n_b, (hipStream_t)stream);
}

int inr_rescale2d_linear(float* out, const float* in, int n_images, int height, int width, int out_height, int out_width,
                         void* stream) {
    INR_REQUIRE(out && in, INR_E_INVALID, "inr_rescale2d_linear: null pointer");
    INR_REQUIRE(n_images >= 0 && height >= 1 && width >= 1 && out_height >= 1 && out_width >= 1, INR_E_INVALID,
                "inr_rescale2d_linear: bad sizes");
    INR_REQUIRE((long long)height * width < (1ll << 31) && (long long)out_height * out_width < (1ll << 31), INR_E_INVALID,
                "inr_rescale2d_linear: image too large");
    return launch_rescale_linear(out, in, n_images, height, width, out_height, out_width, (hipStream_t)stream);
}

int inr_auto_erd(float* accept, const double* values, const float* erd_map, int64_t n_pixels, int n_acquisitions, int rule,
                 void* stream) {
    INR_REQUIRE(accept && values, INR_E_INVALID, "inr_auto_erd: null pointer");
    INR_REQUIRE(n_pixels >= 0 && n_pixels < (1ll << 37), INR_E_INVALID, "inr_auto_erd: bad pixel count");
    return launch_auto_erd(accept, values, erd_map, n_pixels, n_acquisitions, rule, (hipStream_t)stream);
}

// ---- RAMS ------------------------------------------------------------------------------------------------
static int check_rams(const inr_rams_desc_t* d) {
    INR_REQUIRE(d != nullptr, INR_E_INVALID, "rams descriptor is null");
    INR_REQUIRE(d->filters == 32 && d->kernel_size == 3, INR_E_INVALID,
                "rams kernels need filters == 32 and kernel_size == 3 (got %d, %d)", d->filters, d->kernel_size);
    INR_REQUIRE(d->scale >= 1 && d->scale * d->scale <= 32 && d->r >= 1 && d->filters / d->r >= 1 &&
                    d->filters / d->r <= 8 && d->n_rfab >= 0,
                INR_E_INVALID, "bad rams descriptor (scale=%d r=%d N=%d)", d->scale, d->r, d->n_rfab);
    INR_REQUIRE(d->channels >= 3 && d->channels <= 32 && d->channels - 2 * (d->channels / 3) == 3, INR_E_INVALID,
                "rams: channels must leave a temporal depth of 3 before the head (channels=%d)", d->channels);
    return 0;
}

int64_t inr_rams_param_count(const inr_rams_desc_t* desc) {
    if (check_rams(desc)) return INR_E_INVALID;
    return rams_param_floats(desc);
}

size_t inr_rams_workspace_bytes(const inr_rams_desc_t* desc, int batch, int height, int width) {
    if (check_rams(desc) || batch < 1 || height < 3 || width < 3) return 0;
    return rams_workspace_floats(desc, batch, height, width) * sizeof(float);
}

int inr_rams_forward(const inr_rams_desc_t* desc, const float* params, const float* x, float* out, int batch, int height,
                     int width, int clip_round, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_rams(desc)) return rc;
    INR_REQUIRE(params && x && out, INR_E_INVALID, "inr_rams_forward: null pointer");
    INR_REQUIRE(batch >= 1 && batch <= 65535 && height >= 3 && width >= 3, INR_E_INVALID,
                "inr_rams_forward: bad sizes (B=%d H=%d W=%d)", batch, height, width);
    INR_REQUIRE((long long)(height + 4) * (width + 4) * desc->channels * 32 * 4 < (1ll << 31), INR_E_INVALID,
                "inr_rams_forward: one image's activations must stay below 2 GiB");
    INR_REQUIRE(workspace && workspace_bytes >= inr_rams_workspace_bytes(desc, batch, height, width), INR_E_WORKSPACE,
                "inr_rams_forward: workspace too small");
    INR_REQUIRE(aligned16(workspace) && aligned16(params), INR_E_ALIGN, "inr_rams_forward: params/workspace alignment");
    return rams_forward_impl(desc, params, x, out, batch, height, width, clip_round, (float*)workspace,
                             (hipStream_t)stream);
}

// ---- profiler --------------------------------------------------------------------------------------------
int inr_prof_enable(int enable) {
    g_prof.enabled = enable != 0;
    return 0;
}

int inr_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int k = 0; k < KC_COUNT; ++k) {
        for (auto& sp : g_prof.spans[k]) {
            g_prof.pool.push_back(sp.first);
            g_prof.pool.push_back(sp.second);
        }
        g_prof.spans[k].clear();
    }
    return 0;
}

int inr_prof_read(int kernel_class, int64_t* launches, double* total_ms) {
    INR_REQUIRE(kernel_class >= 0 && kernel_class < KC_COUNT && launches && total_ms, INR_E_INVALID,
                "inr_prof_read: bad arguments");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0.0;
    for (auto& sp : g_prof.spans[kernel_class]) {
        INR_HIP(hipEventSynchronize(sp.second));
        float t = 0.f;
        INR_HIP(hipEventElapsedTime(&t, sp.first, sp.second));
        ms += t;
    }
    *launches = (int64_t)g_prof.spans[kernel_class].size();
    *total_ms = ms;
    return 0;
}

// ---- a-15 / (f)-4: RAMS training step ------------------------------------------------------------------------------------------
static int check_rams_train(const inr_rams_desc_t* desc, int B, int H, int W) {
    INR_REQUIRE(desc != nullptr, INR_E_INVALID, "rams descriptor is null");
    INR_REQUIRE(desc->filters == 32 && desc->kernel_size == 3 && desc->channels == 9 && desc->scale >= 1 && desc->scale <= 5 &&
                    desc->r >= 1 && desc->n_rfab >= 0,
                INR_E_INVALID, "rams train: supported configuration is filters 32, kernel 3, channels 9");
    INR_REQUIRE(B >= 1 && H >= 3 && W >= 3 && H == W, INR_E_INVALID,
                "rams train: batch >= 1 and square low-resolution patches (the loss of utils/loss.py works on squares)");
    return 0;
}

int64_t inr_rams_train_param_count(const inr_rams_desc_t* desc) {
    if (!desc) return INR_E_INVALID;
    return rams_train_param_floats(desc);
}

int inr_rams_train_param_offsets(const inr_rams_desc_t* desc, int64_t* offsets, int max_layers) {
    INR_REQUIRE(desc && offsets, INR_E_INVALID, "inr_rams_train_param_offsets: null pointer");
    return rams_train_param_offsets(desc, offsets, max_layers);
}

size_t inr_rams_train_workspace_bytes(const inr_rams_desc_t* desc, int batch, int height, int width) {
    if (!desc || batch < 1 || height < 3 || width < 3) return 0;
    return rams_train_workspace_floats(desc, batch, height, width) * sizeof(float);
}

int inr_rams_train_grads(const inr_rams_desc_t* desc, const float* params, float* grads, const float* x, const float* y_true,
                         const float* mask, double* loss, float* pred, int batch, int height, int width, void* workspace,
                         size_t workspace_bytes, void* stream) {
    if (int rc = check_rams_train(desc, batch, height, width)) return rc;
    INR_REQUIRE(params && grads && x && y_true && mask && loss, INR_E_INVALID, "inr_rams_train_grads: null pointer");
    INR_REQUIRE(workspace && workspace_bytes >= inr_rams_train_workspace_bytes(desc, batch, height, width), INR_E_WORKSPACE,
                "inr_rams_train_grads: workspace too small");
    INR_REQUIRE(aligned16(workspace) && aligned16(params) && aligned16(grads), INR_E_ALIGN,
                "inr_rams_train_grads: params / grads / workspace must be 16-byte aligned");
    return rams_train_grads(desc, params, grads, x, y_true, mask, loss, pred, batch, height, width, (float*)workspace,
                            (hipStream_t)stream);
}

int inr_rams_train_step(const inr_rams_desc_t* desc, float* params, float* grads, float* m, float* v, const float* x,
                        const float* y_true, const float* mask, double* loss, int batch, int height, int width, int64_t step,
                        double lr, double beta1, double beta2, double eps, void* workspace, size_t workspace_bytes, void* stream) {
    INR_REQUIRE(m && v && step >= 1, INR_E_INVALID, "inr_rams_train_step: Adam state missing or step < 1");
    if (int rc = inr_rams_train_grads(desc, params, grads, x, y_true, mask, loss, nullptr, batch, height, width, workspace,
                                      workspace_bytes, stream))
        return rc;
    // Keras Adam: p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)  ==  the kernel's form with eps / sqrt(1 - b2^t)
    const double bc2 = 1.0 - pow(beta2, (double)step);
    return launch_adam(params, grads, m, v, rams_train_param_floats(desc), step, lr, beta1, beta2, eps / sqrt(bc2),
                       (hipStream_t)stream);
}

int inr_debug_set_ptr(int key, void* ptr) {
    if (key == 0) { g_stamps = (unsigned long long*)ptr; return 0; }
    if (key == 1) { g_h3_scratch = (char*)ptr; return 0; }
    return INR_E_INVALID;
}

// ---- diagnostic switches ------------------------------------------------------------------------------------------------
// One table: key -> switch, default, accepted range.  Process-global (see the header): every value is an atomic, so a switch can
// be read while another thread sets it, but a set is visible to every thread's next launch.
namespace {
struct DebugKey {
    int key;
    tune_int* var;
    int def, lo, hi;
};
tune_int g_rams_mode{2}, g_hybrid_mirror{1};   // (keys 14 and 2 fan out to other variables: kept here for inr_debug_get)
const DebugKey* debug_table(int* count) {
    static const DebugKey table[] = {
        {0, &g_force_generic, 0, 0, 1},   {1, &g_mfma16, 1, 0, 1},          {2, &g_hybrid_mirror, 1, 0, 1},
        {3, &g_h3, 1, 0, 2},              {5, &g_h3_serpentine, 1, 0, 1},   {6, &g_h3_wide, 1, 0, 1},
        {7, &g_hp, 1, 0, 1},              {8, &g_stamp_class, -1, -1, 3},   {9, &g_stamp_nth, 0, 0, 1 << 30},
        {10, &g_hp_persistent, 2, 0, 2},  {11, &g_hp_stagger, 0, 0, 1 << 20}, {12, &g_small_multi, 1, 0, 1},
        {13, &g_small_rows, 0, 0, 64},    {14, &g_rams_mode, 2, 0, 7},      {15, &g_rams_lds_waves, 42, 4, 42},
        {16, &g_hp_zhead, 1, 0, 1},       {17, &g_small_spin_limit, 0, 0, 1 << 30}, {18, &g_hp_narrow, 1, 0, 1},
        {19, &g_hp_fused_fwd, 0, 0, 1},  {20, &g_hp_side_stream, 1, 0, 1},  {21, &g_hp_head_min_rows, 16, 4, 256},
        {22, &g_hp_merge_blocks, 256, 28, 1024}, {23, &g_hp_head_rows, 0, 0, 4096},
        {24, &g_rams_epi_fuse, 2, 0, 2}, {25, &g_reduce_onepass, 1, 0, 1},
        {26, &g_rams_pregate_min_vox, 600000, 0, 1 << 30},
        {27, &g_hp_row, 0, 0, 1},        {28, &g_hp_row_min_tiles, 1024, 1, 1 << 30},
        {29, &g_hp_narrow_max_tiles, 192, 0, 1 << 30},
        {30, &g_hp_row_head, 1, 0, 1},   {31, &g_hp_row_head_min_tiles, 768, 1, 1 << 30},
    };
    *count = (int)(sizeof(table) / sizeof(table[0]));
    return table;
}
void debug_apply(const DebugKey& k, int value) {
    if (k.key == 14) {
        g_rams_h3 = value & 3;
        g_rams_force_lds = (value >> 2) & 1;
    } else if (k.key == 15) {
        value = (value == 8 || value == 16 || value == 42) ? value : 4;
    } else if (k.key == 2) {
        set_hybrid_variant(value);
    }
    k.var->store(value);
}
}  // namespace

int inr_debug_set(int key, int value) {
    int n = 0;
    const DebugKey* t = debug_table(&n);
    for (int i = 0; i < n; ++i)
        if (t[i].key == key) {
            INR_REQUIRE(value >= t[i].lo && value <= t[i].hi, INR_E_INVALID, "inr_debug_set: key %d takes %d .. %d (got %d)", key,
                        t[i].lo, t[i].hi, value);
            debug_apply(t[i], value);
            return 0;
        }
    INR_REQUIRE(false, INR_E_INVALID, "inr_debug_set: unknown key %d", key);
}

int inr_debug_get(int key, int* value) {
    INR_REQUIRE(value != nullptr, INR_E_INVALID, "inr_debug_get: null pointer");
    int n = 0;
    const DebugKey* t = debug_table(&n);
    for (int i = 0; i < n; ++i)
        if (t[i].key == key) {
            *value = t[i].var->load();
            return 0;
        }
    INR_REQUIRE(false, INR_E_INVALID, "inr_debug_get: unknown key %d", key);
}

int inr_debug_reset(void) {
    int n = 0;
    const DebugKey* t = debug_table(&n);
    for (int i = 0; i < n; ++i) debug_apply(t[i], t[i].def);
    g_stamps = nullptr;
    g_h3_scratch = nullptr;
    return 0;
}

int inr_launch_count(int family, int64_t* count) {
    INR_REQUIRE(family >= 0 && family < LF_COUNT && count, INR_E_INVALID, "inr_launch_count: bad arguments");
    *count = (int64_t)g_launches[family].load(std::memory_order_relaxed);
    return 0;
}

int inr_launch_counts_reset(void) {
    for (int f = 0; f < LF_COUNT; ++f) g_launches[f].store(0, std::memory_order_relaxed);
    return 0;
}

int inr_sincos_probe(float* s, float* c, const float* x, int64_t n, void* stream) {
    INR_REQUIRE(s && c && x && n >= 0, INR_E_INVALID, "inr_sincos_probe: bad arguments");
    return launch_sincos_probe(s, c, x, n, (hipStream_t)stream);
}

}  // extern "C"
