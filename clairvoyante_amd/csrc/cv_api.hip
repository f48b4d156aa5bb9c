// cv_api.hip -- the C ABI (include/clairvoyante_amd.h): model object, parameter
// table, chunked forward.  Each entry point cites the reference method it replaces
// in the header.
#include "cv_internal.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>

static thread_local char g_err[512] = "";

void cv_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *cv_last_error(void) { return g_err; }

// variable names of the reference graph (jupyter_nb/visualization.ipynb:103-120)
static const char *const kParamNames[CV_NUM_PARAMS] = {
    "conv1/kernel", "conv1/bias", "conv2/kernel", "conv2/bias", "conv3/kernel", "conv3/bias",
    "fc4/kernel", "fc4/bias", "fc5/kernel", "fc5/bias",
    "YBaseChangeSigmoid/kernel", "YBaseChangeSigmoid/bias", "YZygosityFC/kernel", "YZygosityFC/bias",
    "YVarTypeFC/kernel", "YVarTypeFC/bias", "YIndelLengthFC/kernel", "YIndelLengthFC/bias"};

static int compute_shapes(cv_model *m)
{
    const cv_arch &a = m->arch;
    cv_shapes &s = m->sh;
    int h = CV_INPUT_H, c = CV_INPUT_C;
    for (int l = 0; l < 3; l++) {
        if (a.kh[l] < 1 || a.kh[l] > 7 || a.cout[l] < 1 || a.cout[l] > 64 || a.pool[l] < 1 ||
            a.pool[l] > h) {
            cv_set_error("cv_create: unsupported layer %d (kh %d cout %d pool %d)", l + 1, a.kh[l],
                         a.cout[l], a.pool[l]);
            return 1;
        }
        s.cin[l] = c;
        s.hc[l] = h;
        h -= a.pool[l] - 1;
        s.hp[l] = h;
        c = a.cout[l];
        s.ntile[l] = (a.cout[l] + 15) / 16;
        s.cinb[l] = (s.cin[l] + 15) / 16;
    }
    if (a.fc4 < 1 || a.fc4 > 512 || a.fc5 < 1 || a.fc5 > 512) {
        cv_set_error("cv_create: unsupported fc sizes %d/%d", a.fc4, a.fc5);
        return 1;
    }
    s.flat = h * 4 * c;
    s.kb4 = h * 4 * s.ntile[2];
    s.nb4 = (a.fc4 + 15) / 16;
    s.nb5 = (a.fc5 + 15) / 16;
    // parameter table
    int p = 0;
    auto add = [&](int nd, int64_t d0, int64_t d1, int64_t d2, int64_t d3) {
        m->pndim[p] = nd;
        m->pdims[p][0] = d0; m->pdims[p][1] = d1; m->pdims[p][2] = d2; m->pdims[p][3] = d3;
        int64_t sz = 1;
        for (int i = 0; i < nd; i++) sz *= m->pdims[p][i];
        m->psize[p] = sz;
        p++;
    };
    for (int l = 0; l < 3; l++) {
        add(4, a.kh[l], 4, s.cin[l], a.cout[l]);
        add(1, a.cout[l], 1, 1, 1);
    }
    add(2, s.flat, a.fc4, 1, 1); add(1, a.fc4, 1, 1, 1);
    add(2, a.fc4, a.fc5, 1, 1);  add(1, a.fc5, 1, 1, 1);
    add(2, a.fc4, 4, 1, 1);      add(1, 4, 1, 1, 1);
    add(2, a.fc5, 2, 1, 1);      add(1, 2, 1, 1, 1);
    add(2, a.fc5, 4, 1, 1);      add(1, 4, 1, 1, 1);
    add(2, a.fc5, 6, 1, 1);      add(1, 6, 1, 1, 1);
    m->poff[0] = 0;
    for (int i = 0; i < CV_NUM_PARAMS; i++) m->poff[i + 1] = m->poff[i] + m->psize[i];
    return 0;
}

static int find_param(const char *name)
{
    for (int i = 0; i < CV_NUM_PARAMS; i++)
        if (strcmp(name, kParamNames[i]) == 0) return i;
    cv_set_error("unknown variable '%s'", name);
    return -1;
}

extern "C" int cv_create(const cv_arch *arch, int device, cv_model **out)
{
    if (!arch || !out) { cv_set_error("cv_create: null argument"); return 1; }
    int ndev = 0;
    CV_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        cv_set_error("cv_create: device %d not present (%d visible)", device, ndev);
        return 1;
    }
    CV_HIP(hipSetDevice(device));
    cv_model *m = new (std::nothrow) cv_model();
    if (!m) { cv_set_error("cv_create: out of host memory"); return 1; }
    memset(m, 0, sizeof(*m));
    m->arch = *arch;
    m->device = device;
    m->impl = 1;
    m->chunk = 65536;
    if (compute_shapes(m)) { delete m; return 1; }
    const int64_t np = m->poff[CV_NUM_PARAMS];
    const cv_shapes &s = m->sh;
    hipError_t e = hipSuccess;
    auto alloc = [&](float **p, size_t nfloat) {
        if (e == hipSuccess) e = hipMalloc(p, sizeof(float) * nfloat);
        if (e == hipSuccess) e = hipMemset(*p, 0, sizeof(float) * nfloat);
    };
    // the 8 doubles of the current pass's losses live right behind the model's own gradient array: the step zeroes both
    // with ONE memset (a launch less at the head of every step)
    const int64_t np4 = (np + 3) / 4 * 4;
    alloc(&m->params, np); alloc(&m->grads_own, np4 + CV_GRAD_HEADER + 16); alloc(&m->adam_m, np); alloc(&m->adam_v, np);
    if (m->grads_own) { m->grads = m->grads_own + CV_GRAD_HEADER; m->loss_dev = reinterpret_cast<double *>(m->grads + np4); }
    m->train_overlap = 1;
    m->train_sides = 3;
    m->train_ksplit = 1;
    m->tiny_g = 400;
    // (rounds 1-4: 160 / 256 / 2 048, set where each kernel was first tuned; a ladder of batch sizes showed steps of 75 us,
    // 105 us and 840 us in the time of a call one group past each line: profiles/r05/infer_size_sweep.txt)
    m->inf_small_g = 256; m->inf_fc4_small_g = 288; m->inf_slab_g = -1; m->inf_flat = 1; m->inf_slim_small_g = -1; m->inf_fc4_one_g = 80;      // (-1: fc4's kernel form by estimate, cv_mfma_forward)
    m->sched = 3839;
    alloc(&m->wp_conv1, 4 * 64);
    for (int l = 1; l < 3; l++) alloc(&m->wp_conv[l], (size_t)s.ntile[l] * arch->kh[l] * 4 * s.cinb[l] * 256);
    alloc(&m->wp_fc4, (size_t)s.kb4 * ((s.nb4 + 3) / 4 * 4) * 256);   // fragments padded to the wave count
    alloc(&m->wp_fc5, (size_t)s.nb4 * ((s.nb5 + 3) / 4 * 4) * 256);
    for (int l = 1; l < 3; l++) alloc(&m->wpd_conv[l], (size_t)s.cinb[l] * arch->kh[l] * 4 * s.ntile[l] * 256);
    alloc(&m->wpd_fc4, (size_t)((s.kb4 + 23) / 24) * s.nb4 * 24 * 256);
    alloc(&m->wpd_fc5, (size_t)s.nb5 * 24 * 256);
    if (s.nb4 == 21 && s.ntile[2] == 3 && arch->pool[2] == 3) alloc(&m->wpr_fc4, (size_t)4 * s.ntile[2] * s.hp[2] * 24 * 256);
    if (s.nb4 == 21) alloc(&m->wps_fc4, (size_t)3 * s.kb4 * 8 * 256);
    if (s.nb4 == 21) alloc(&m->wps7_fc4, (size_t)7 * s.kb4 * 3 * 256);
    if (s.nb4 == 21) alloc(&m->wps21_fc4, (size_t)21 * s.kb4 * 256);
    else if (s.nb4 == 3) alloc(&m->wps7_fc4, (size_t)3 * s.kb4 * 256);          // slim: 3 slabs of one fragment
    if (s.nb4 == 21 && s.nb5 == 11) alloc(&m->wps3_fc5, (size_t)3 * s.nb4 * 4 * 256);
    if (s.nb4 == 21 && s.nb5 == 11) alloc(&m->wp5p_fc5, (size_t)((s.nb4 + 3) / 4) * 48 * 256);
    alloc(&m->wp_heads0, (size_t)s.nb4 * 256);
    alloc(&m->wp_heads1, (size_t)s.nb5 * 256);
    alloc(&m->wp_heads12, (size_t)s.nb5 * 16 * 12 + 16);
    m->variant = 2031;
    if (e == hipSuccess) e = hipMalloc(&m->loss_acc, sizeof(double) * 8);
    if (e == hipSuccess) e = hipMemset(m->loss_acc, 0, sizeof(double) * 8);
    if (e != hipSuccess) {
        cv_set_error("cv_create: device allocation failed: %s", hipGetErrorString(e));
        cv_destroy(m);
        return 1;
    }
    cv_layouts_stale(m);
    *out = m;
    return 0;
}

extern "C" int cv_destroy(cv_model *m)
{
    if (!m) return 0;
    hipSetDevice(m->device);
    float *bufs[] = {m->params, m->grads_own, m->adam_m, m->adam_v, m->wp_conv1, m->wp_conv[1], m->wp_conv[2],
                     m->wp_fc4, m->wp_fc5, m->wp_heads0, m->wp_heads1, m->wp_heads12, m->wpd_conv[1], m->wpd_conv[2], m->wpd_fc4, m->wpr_fc4, m->wpd_fc5, m->wps_fc4, m->wps7_fc4, m->wps21_fc4, m->wps3_fc5, m->wp5p_fc5, m->wg_part, m->tm_p1, m->tm_p2, m->tm_p3, m->tm_h4, m->tm_h5, m->r_a[0],
                     m->r_a[1], m->r_a[2], m->r_p[0], m->r_p[1], m->r_p[2], m->r_h4, m->r_h5, m->t_buf, m->tr_keep};
    for (float *b : bufs)
        if (b) hipFree(b);
    if (m->tail_dev) hipFree(m->tail_dev);
    if (m->loss_acc) hipFree(m->loss_acc);
    if (m->loss_rows) hipFree(m->loss_rows);
    if (m->l2_rows) hipFree(m->l2_rows);
    if (m->tr_side) {
        (void)hipStreamSynchronize(m->tr_side);
        (void)hipStreamDestroy(m->tr_side);
        for (int i = 0; i < 2; i++) if (m->tr_side_more[i]) { (void)hipStreamSynchronize(m->tr_side_more[i]); (void)hipStreamDestroy(m->tr_side_more[i]); }
        for (int i = 0; i < CV_TR_EVENTS; i++) (void)hipEventDestroy(m->tr_ev[i]);
        (void)hipEventDestroy(m->tr_dense_ready);
        (void)hipEventDestroy(m->tr_l2_done);
        (void)hipEventDestroy(m->tr_pack_fork); (void)hipEventDestroy(m->tr_pack_done);
    }
    cv_prof_free(m);
    delete m;
    return 0;
}

extern "C" int cv_param_info(const cv_model *m, int idx, const char **tf_name, int *ndim, int64_t dims[4])
{
    if (!m || idx < 0 || idx >= CV_NUM_PARAMS) { cv_set_error("cv_param_info: bad index %d", idx); return 1; }
    if (tf_name) *tf_name = kParamNames[idx];
    if (ndim) *ndim = m->pndim[idx];
    if (dims) for (int i = 0; i < 4; i++) dims[i] = m->pdims[idx][i];
    return 0;
}

extern "C" int cv_param_buffer(cv_model *m, float **flat_dev, int64_t *count, int64_t *offsets)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (flat_dev) *flat_dev = m->params;
    if (count) *count = m->poff[CV_NUM_PARAMS];
    if (offsets) for (int i = 0; i <= CV_NUM_PARAMS; i++) offsets[i] = m->poff[i];
    return 0;
}

extern "C" int cv_set_param(cv_model *m, const char *tf_name, const float *src, int64_t count, void *stream)
{
    if (!m || !tf_name || !src) { cv_set_error("cv_set_param: null argument"); return 1; }
    int i = find_param(tf_name);
    if (i < 0) return 1;
    if (count != m->psize[i]) {
        cv_set_error("cv_set_param(%s): got %lld values, variable holds %lld", tf_name, (long long)count,
                     (long long)m->psize[i]);
        return 1;
    }
    CV_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    CV_HIP(hipMemcpyAsync(m->params + m->poff[i], src, sizeof(float) * count, hipMemcpyHostToDevice, st));
    CV_HIP(hipStreamSynchronize(st));
    cv_layouts_stale(m);
    return 0;
}

extern "C" int cv_get_param(cv_model *m, const char *tf_name, float *dst, int64_t count, void *stream)
{
    if (!m || !tf_name || !dst) { cv_set_error("cv_get_param: null argument"); return 1; }
    int i = find_param(tf_name);
    if (i < 0) return 1;
    if (count != m->psize[i]) {
        cv_set_error("cv_get_param(%s): asked %lld values, variable holds %lld", tf_name, (long long)count,
                     (long long)m->psize[i]);
        return 1;
    }
    CV_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    CV_HIP(hipMemcpyAsync(dst, m->params + m->poff[i], sizeof(float) * count, hipMemcpyDeviceToHost, st));
    CV_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int cv_params_changed(cv_model *m)
{
    if (!m) { cv_set_error("null model"); return 1; }
    cv_layouts_stale(m);
    return 0;
}

extern "C" int cv_set_option(cv_model *m, const char *key, int64_t value)
{
    if (!m || !key) { cv_set_error("cv_set_option: null argument"); return 1; }
    if (!strcmp(key, "impl")) {
        if (value != 0 && value != 1) { cv_set_error("impl must be 0 or 1"); return 1; }
        m->impl = (int)value;
        return 0;
    }
    if (!strcmp(key, "profile")) { m->profile = value ? 1 : 0; return 0; }
    if (!strcmp(key, "keep_activations")) { m->keep_act = value ? 1 : 0; return 0; }
    if (!strcmp(key, "train_overlap")) { m->train_overlap = value ? 1 : 0; return 0; }
    if (!strcmp(key, "train_ksplit")) { m->train_ksplit = value ? 1 : 0; return 0; }
    if (!strcmp(key, "train_side_streams")) { m->train_sides = value < 1 ? 1 : (value > 3 ? 3 : (int)value); return 0; }
    if (!strcmp(key, "infer_small_groups")) { m->inf_small_g = value < 0 ? 0 : (int)(value > 65536 ? 65536 : value); return 0; }
    if (!strcmp(key, "infer_fc4_small_groups")) { m->inf_fc4_small_g = value < 0 ? 0 : (int)(value > 65536 ? 65536 : value); return 0; }
    if (!strcmp(key, "infer_fc4_one_groups")) { m->inf_fc4_one_g = value < 0 ? 0 : (int)(value > 65536 ? 65536 : value); return 0; }
    if (!strcmp(key, "infer_slab_groups")) { m->inf_slab_g = value < 0 ? -1 : (int)(value > 65536 ? 65536 : value); return 0; }
    if (!strcmp(key, "infer_flat")) { m->inf_flat = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return 0; }
    if (!strcmp(key, "slim_waves")) { m->inf_slim_waves = (value == 4 || value == 8) ? (int)value : 0; return 0; }
    if (!strcmp(key, "slim_small_groups")) { m->inf_slim_small_g = value < 0 ? -1 : (int)(value > 65536 ? 65536 : value); return 0; }
    if (!strcmp(key, "dense_rag")) { m->inf_rag_s = value < 0 ? -1 : (value > 14 ? 14 : (int)value); return 0; }
    if (!strcmp(key, "train_tiny_groups")) { m->tiny_g = value < 0 ? 0 : (value > 4096 ? 4096 : (int)value); return 0; }
    if (!strcmp(key, "variant")) { m->variant = (int)value; return 0; }
    if (!strcmp(key, "train_sched")) { m->sched = (int)value & 8191; return 0; }
    if (!strncmp(key, "dbg", 3) && key[3] >= '0' && key[3] <= '7' && !key[4]) { m->dbg[key[3] - '0'] = (int)value; cv_layouts_stale(m, CVL_BACKWARD); return 0; }
    if (!strcmp(key, "chunk")) {
        if (value < 16 || value > (1 << 22)) { cv_set_error("chunk must be in [16, 4194304]"); return 1; }
        m->chunk = (value + 15) / 16 * 16;
        return 0;
    }
    cv_set_error("unknown option '%s'", key);
    return 1;
}

extern "C" int cv_get_option(const cv_model *m, const char *key, int64_t *value)
{
    if (!m || !key || !value) { cv_set_error("cv_get_option: null argument"); return 1; }
    if (!strcmp(key, "impl")) { *value = m->impl; return 0; }
    if (!strcmp(key, "chunk")) { *value = m->chunk; return 0; }
    if (!strcmp(key, "profile")) { *value = m->profile; return 0; }
    if (!strcmp(key, "keep_activations")) { *value = m->keep_act; return 0; }
    if (!strcmp(key, "train_overlap")) { *value = m->train_overlap; return 0; }
    if (!strcmp(key, "train_ksplit")) { *value = m->train_ksplit; return 0; }
    if (!strcmp(key, "train_side_streams")) { *value = m->train_sides; return 0; }
    if (!strcmp(key, "infer_small_groups")) { *value = m->inf_small_g; return 0; }
    if (!strcmp(key, "infer_fc4_small_groups")) { *value = m->inf_fc4_small_g; return 0; }
    if (!strcmp(key, "infer_fc4_one_groups")) { *value = m->inf_fc4_one_g; return 0; }
    if (!strcmp(key, "infer_slab_groups")) { *value = m->inf_slab_g; return 0; }
    if (!strcmp(key, "infer_flat")) { *value = m->inf_flat; return 0; }
    if (!strcmp(key, "slim_waves")) { *value = m->inf_slim_waves; return 0; }
    if (!strcmp(key, "slim_small_groups")) { *value = m->inf_slim_small_g; return 0; }
    if (!strcmp(key, "dense_rag")) { *value = m->inf_rag_s; return 0; }
    if (!strcmp(key, "train_tiny_groups")) { *value = m->tiny_g; return 0; }
    if (!strcmp(key, "train_sched")) { *value = m->sched; return 0; }
    if (!strcmp(key, "variant")) { *value = m->variant; return 0; }
    if (!strncmp(key, "dbg", 3) && key[3] >= '0' && key[3] <= '7' && !key[4]) { *value = m->dbg[key[3] - '0']; return 0; }
    cv_set_error("unknown option '%s'", key);
    return 1;
}

extern "C" int cv_forward(cv_model *m, const float *x_dev, int64_t n, float *out16_dev, void *stream)
{
    if (!m) { cv_set_error("cv_forward: null model"); return 1; }
    if (n < 0) { cv_set_error("cv_forward: negative batch"); return 1; }
    if (n == 0) return 0;    // the reference runs predict on an empty final batch (utils_v2.py:56-59)
    if (!x_dev || !out16_dev) { cv_set_error("cv_forward: null buffer"); return 1; }
    CV_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t chunk = m->impl ? m->chunk : (m->chunk < 4096 ? m->chunk : 4096);
    for (int64_t off = 0; off < n; off += chunk) {
        int64_t cn = n - off < chunk ? n - off : chunk;
        const float *xc = x_dev + (size_t)off * (CV_INPUT_H * CV_INPUT_W * CV_INPUT_C);
        float *oc = out16_dev + (size_t)off * CV_NUM_OUT;
        int rc = m->impl ? cv_mfma_forward(m, xc, cn, oc, st) : cv_ref_forward(m, xc, cn, oc, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cv_get_activation(cv_model *m, int layer, float *dst_dev, int64_t n, void *stream)
{
    if (!m || !dst_dev) { cv_set_error("cv_get_activation: null argument"); return 1; }
    if ((layer >= 11 && layer <= 13) || (layer >= 21 && layer <= 23)) {   // maps of the last training slice (single-slice steps)
        const int l = layer % 10 - 1; const bool grad = layer > 20;
        const float *src = grad ? m->last_tr_gpre[l] : m->last_tr_pool[l];
        if (n <= 0 || n > m->last_tr_n || !src) {
            cv_set_error("cv_get_activation: layer %d of the last training slice (%lld candidates) is not there", layer, (long long)m->last_tr_n);
            return 1;
        }
        CV_HIP(hipSetDevice(m->device));
        const int npos = (grad ? m->sh.hc[l] : m->sh.hp[l]) * 4;
        if (m->last_tr_tile) return cv_tm_to_natural(src, npos * m->sh.ntile[l], m->sh.ntile[l] * 16, m->arch.cout[l], npos, n, dst_dev, (hipStream_t)stream);
        CV_HIP(hipMemcpyAsync(dst_dev, src, sizeof(float) * (size_t)npos * m->arch.cout[l] * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return 0;
    }
    if (layer < 1 || layer > 7) { cv_set_error("cv_get_activation: layer %d not in 1..7, 11..13, 21..23", layer); return 1; }
    if (layer >= 6) {        // training-pass tensors: 6 = keep mask scaled by a (0 where dropped), 7 = dropout output
        if (n <= 0 || n > m->last_tr_n || !m->last_tr_d4) {
            cv_set_error("cv_get_activation: n=%lld but the last training slice held %lld candidates", (long long)n,
                         (long long)m->last_tr_n);
            return 1;
        }
        CV_HIP(hipSetDevice(m->device));
        const float *src = layer == 6 ? m->last_tr_mask : m->last_tr_d4;
        if (m->last_tr_tile) return cv_tm_to_natural(src, m->sh.nb4, m->sh.nb4 * 16, m->arch.fc4, 1, n, dst_dev, (hipStream_t)stream);
        CV_HIP(hipMemcpyAsync(dst_dev, src, sizeof(float) * (size_t)m->arch.fc4 * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return 0;
    }
    if (n <= 0 || n > m->last_n) {
        cv_set_error("cv_get_activation: n=%lld but the last pass held %lld candidates", (long long)n,
                     (long long)m->last_n);
        return 1;
    }
    CV_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const cv_shapes &s = m->sh;
    const cv_arch &a = m->arch;
    if (m->last_impl == 0) {
        const float *src; size_t per;
        if (layer <= 3) { src = m->r_p[layer - 1]; per = (size_t)s.hp[layer - 1] * 4 * a.cout[layer - 1]; }
        else if (layer == 4) { src = m->r_h4; per = a.fc4; }
        else { src = m->r_h5; per = a.fc5; }
        CV_HIP(hipMemcpyAsync(dst_dev, src, sizeof(float) * per * n, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (layer == 1 && m->last_impl == 1 && (m->last_variant & 1) && !m->stage_kernel[0]) {
        cv_set_error("layer 1 is not materialised while the first layer is fused into the conv2 kernel "
                     "(option variant bit 0); clear the bit to inspect it");
        return 1;
    }
    if (layer == 3 && m->last_impl == 1 && m->stage_kernel[2] && !m->stage_kernel[3]) {
        cv_set_error("layer 3 is not materialised while conv3 and fc4 run as one kernel (option variant bit 8); "
                     "clear the bit to inspect it");
        return 1;
    }
    if (layer >= 4 && m->last_impl == 1 && !m->last_maps) {
        cv_set_error("layers 4 / 5 are not materialised while fc5 and the heads ride on the fc4 kernel (option variant bit 10) "
                     "unless option keep_activations is set before the pass");
        return 1;
    }
    if (layer <= 3) {
        int l = layer - 1;
        const float *tm = l == 0 ? m->tm_p1 : (l == 1 ? m->tm_p2 : m->tm_p3);
        int npos = s.hp[l] * 4;
        return cv_tm_to_natural(tm, npos * s.ntile[l], s.ntile[l] * 16, a.cout[l], npos, n, dst_dev, st);
    }
    if (layer == 4) return cv_tm_to_natural(m->tm_h4, s.nb4, s.nb4 * 16, a.fc4, 1, n, dst_dev, st);
    return cv_tm_to_natural(m->tm_h5, s.nb5, s.nb5 * 16, a.fc5, 1, n, dst_dev, st);
}

// ---- per-kernel timing ---------------------------------------------------------
#include <vector>
struct cv_prof {
    struct rec { hipEvent_t a, b; int stage; };
    std::vector<rec> recs;        // events in flight
    std::vector<hipEvent_t> pool; // recycled events
    hipEvent_t cur[CV_NUM_STAGES];
    double ms[CV_NUM_STAGES];
    int64_t cnt[CV_NUM_STAGES];
};

static hipEvent_t prof_event(cv_prof *p)
{
    if (!p->pool.empty()) { hipEvent_t e = p->pool.back(); p->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void cv_prof_begin(cv_model *m, int stage, hipStream_t st)
{
    if (!m->profile) return;
    if (!m->prof) { cv_prof *p = new cv_prof(); for (int i = 0; i < CV_NUM_STAGES; i++) { p->ms[i] = 0; p->cnt[i] = 0; } m->prof = p; }
    cv_prof *p = (cv_prof *)m->prof;
    p->cur[stage] = prof_event(p);
    (void)hipEventRecord(p->cur[stage], st);
}

void cv_prof_end(cv_model *m, int stage, hipStream_t st)
{
    if (!m->profile || !m->prof) return;
    cv_prof *p = (cv_prof *)m->prof;
    hipEvent_t b = prof_event(p);
    (void)hipEventRecord(b, st);
    p->recs.push_back({p->cur[stage], b, stage});
}

void cv_prof_free(cv_model *m)
{
    cv_prof *p = (cv_prof *)m->prof;
    if (!p) return;
    for (auto &r : p->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : p->pool) (void)hipEventDestroy(e);
    delete p;
    m->prof = nullptr;
}

extern "C" int cv_kernel_name(const cv_model *m, int stage, const char **name)
{
    if (!m || !name || stage < 0 || stage >= CV_NUM_STAGES) { cv_set_error("cv_kernel_name: bad argument"); return 1; }
    *name = m->last_impl == 1 ? m->stage_kernel[stage] : nullptr;
    return 0;
}

extern "C" int cv_kernel_times(cv_model *m, double ms[CV_NUM_STAGES], int64_t launches[CV_NUM_STAGES])
{
    if (!m || !ms || !launches) { cv_set_error("cv_kernel_times: null argument"); return 1; }
    for (int i = 0; i < CV_NUM_STAGES; i++) { ms[i] = 0; launches[i] = 0; }
    cv_prof *p = (cv_prof *)m->prof;
    if (!p) return 0;
    CV_HIP(hipSetDevice(m->device));
    for (auto &r : p->recs) {
        CV_HIP(hipEventSynchronize(r.b));
        float t = 0;
        CV_HIP(hipEventElapsedTime(&t, r.a, r.b));
        p->ms[r.stage] += t; p->cnt[r.stage]++;
        p->pool.push_back(r.a); p->pool.push_back(r.b);
    }
    p->recs.clear();
    for (int i = 0; i < CV_NUM_STAGES; i++) { ms[i] = p->ms[i]; launches[i] = p->cnt[i]; p->ms[i] = 0; p->cnt[i] = 0; }
    return 0;
}
