/*
 * velox_oracle.c — CPU restatement of the Velox frame-codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under velox_amd/ (the product) may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
 * as the checker.  Parity of this restatement is PINNED: tests/test_oracle_vs_reference.py runs
 * it against oracle/_ref/libvelox_ref.so (the real reference compiled from /root/reference by
 * oracle/Makefile) and tests/test_oracle_golden.py against the committed tests/golden/ vectors
 * that were generated from that same reference build.
 *
 * Written from scratch in plain C as a flat, batch-oriented codec (arrays of streams, a word-wide
 * LSB-first bit accumulator) — it follows the reference's *algorithm*, each function citing the
 * reference lines it restates (paths relative to /root/reference), not its code structure.
 *
 * Build: gcc -O2 -fwrapv -ffp-contract=off -fPIC -shared -pthread   (see oracle/Makefile)
 *   -ffp-contract=off : ComputeLPC is IEEE double without FMA in the reference build.
 *   -fwrapv           : int32/int64 overflow wraps, as it does in the compiled reference.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

#define VLX_SUB_BLOCK   8192   /* VeloxCore.h:239 */
#define VLX_LPC_ORDER   8      /* VeloxCore.h:88,136 */
#define VLX_NEURAL_TAPS 12     /* VeloxCore.h:17 */

/* ------------------------------------------------------------------ bit IO */
/* LSB-first bit writer; restates BitStreamWriter (VeloxEntropy.h:9-43) with a 64-bit word
 * accumulator instead of per-bit pushes.  Bytes come out in the same order. */
typedef struct {
    uint8_t *buf;
    size_t   cap;
    size_t   len;     /* whole bytes emitted */
    uint64_t acc;
    int      cnt;     /* pending bits in acc (0..7 after drain) */
} bitw;

static void bw_init(bitw *w) { w->buf = NULL; w->cap = 0; w->len = 0; w->acc = 0; w->cnt = 0; }
static void bw_free(bitw *w) { free(w->buf); w->buf = NULL; }
static void bw_reserve(bitw *w, size_t more)
{
    if (w->len + more <= w->cap) return;
    size_t nc = w->cap ? w->cap * 2 : 1024;
    while (nc < w->len + more) nc *= 2;
    w->buf = (uint8_t *)realloc(w->buf, nc);
    w->cap = nc;
}
static void bw_drain(bitw *w)
{
    bw_reserve(w, 8);
    while (w->cnt >= 8) { w->buf[w->len++] = (uint8_t)w->acc; w->acc >>= 8; w->cnt -= 8; }
}
/* Write the low n bits of v, LSB first (VeloxEntropy.h:31-35).  n in [0,64]. */
static void bw_write(bitw *w, uint64_t v, int n)
{
    while (n > 0) {
        int take = n > 32 ? 32 : n;
        uint64_t part = (take == 64) ? v : (v & ((1ULL << take) - 1));
        w->acc |= part << w->cnt;       /* cnt <= 7, take <= 32: fits */
        w->cnt += take;
        bw_drain(w);
        v >>= take; n -= take;
    }
}
static void bw_ones(bitw *w, uint64_t q)
{
    while (q >= 32) { bw_write(w, 0xFFFFFFFFu, 32); q -= 32; }
    if (q) bw_write(w, (1ULL << q) - 1, (int)q);
}
/* Flush: pad the last partial byte with zero bits (VeloxEntropy.h:37-41). */
static void bw_flush(bitw *w)
{
    if (w->cnt > 0) { bw_reserve(w, 1); w->buf[w->len++] = (uint8_t)w->acc; w->acc = 0; w->cnt = 0; }
}
static size_t bw_bits(const bitw *w) { return w->len * 8 + (size_t)w->cnt; }

/* LSB-first bit reader; reads past the end yield zero bits (VeloxEntropy.h:57-70). */
typedef struct { const uint8_t *d; size_t size; size_t pos; /* bit position */ } bitr;

static void br_init(bitr *r, const uint8_t *d, size_t size) { r->d = d; r->size = size; r->pos = 0; }
static inline int br_bit(bitr *r)
{
    size_t byte = r->pos >> 3;
    if (byte >= r->size) return 0;      /* the reference does not advance either; harmless */
    int b = (r->d[byte] >> (r->pos & 7)) & 1;
    r->pos++;
    return b;
}
static uint64_t br_read(bitr *r, int n)                     /* VeloxEntropy.h:73-80 */
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) if (br_bit(r)) v |= 1ULL << i;
    return v;
}
static int64_t br_read_s(bitr *r, int n)                    /* VeloxEntropy.h:82-88 */
{
    uint64_t v = br_read(r, n);
    if (v & (1ULL << (n - 1))) return (int64_t)(v - (1ULL << n));
    return (int64_t)v;
}

/* ------------------------------------------------------------- entropy code */
static inline uint64_t zigzag(int64_t n) { return ((uint64_t)n << 1) ^ (uint64_t)(n >> 63); }   /* VeloxEntropy.h:95 */
static inline int64_t dezigzag(uint64_t n) { return (int64_t)((n >> 1) ^ (uint64_t)(-(int64_t)(n & 1))); } /* :96 */

/* Adaptive Rice code with escape (VeloxEntropy.h:98-119). */
static void rice_put(bitw *w, int64_t val, int k)
{
    uint64_t m = zigzag(val);
    uint64_t q = m >> k;
    if (q < 64) {
        bw_ones(w, q);
        bw_write(w, 0, 1);
        if (k > 0) bw_write(w, m & ((1ULL << k) - 1), k);
    } else {
        bw_ones(w, 64);
        bw_write(w, 0, 1);
        bw_write(w, m, 40);
    }
}
static int64_t rice_get(bitr *r, int k)                      /* VeloxEntropy.h:121-138 */
{
    uint64_t q = 0;
    while (br_bit(r)) q++;
    uint64_t m;
    if (q < 64) {
        uint64_t rem = (k > 0) ? br_read(r, k) : 0;
        m = (q << k) | rem;
    } else {
        m = br_read(r, 40);
    }
    return dezigzag(m);
}

static inline int rice_k(uint64_t run_avg)                   /* VeloxCore.h:109-114 */
{
    if (run_avg == 0) return 0;
    return 63 - __builtin_clzll(run_avg);
}
static inline uint64_t run_avg_next(uint64_t ra, uint64_t m) /* VeloxCore.h:120-121 */
{
    ra = ra - (ra >> 3) + (m >> 3);
    return ra < 1 ? 1 : ra;
}

/* ------------------------------------------------- sign-sign LMS ("neural") */
typedef struct { int32_t w[VLX_NEURAL_TAPS]; int32_t h[VLX_NEURAL_TAPS]; } lms12;

static void lms_init(lms12 *p) { memset(p, 0, sizeof(*p)); }
static inline int32_t lms_predict(const lms12 *p)            /* VeloxCore.h:22-26 */
{
    int64_t s = 0;
    for (int i = 0; i < VLX_NEURAL_TAPS; i++) s += (int64_t)p->h[i] * p->w[i];
    return (int32_t)(s >> 11);
}
static inline void lms_update(lms12 *p, int32_t actual, int32_t pred)   /* VeloxCore.h:27-40 */
{
    int32_t err = (int32_t)((uint32_t)actual - (uint32_t)pred);
    if (err == 0) return;                       /* note: history is NOT shifted (:30) */
    int sign = err > 0 ? 1 : -1;
    int32_t aerr = err < 0 ? (int32_t)(0u - (uint32_t)err) : err;   /* abs(INT_MIN) stays INT_MIN */
    int delta = (aerr > 1024) ? 16 : 4;
    for (int i = 0; i < VLX_NEURAL_TAPS; i++) {
        int hs = (p->h[i] > 0) - (p->h[i] < 0);
        if (hs == sign) p->w[i] += delta;
        else if (hs != 0) p->w[i] -= delta;
        if ((i & 7) == 0) { if (p->w[i] > 0) p->w[i]--; if (p->w[i] < 0) p->w[i]++; }
    }
    memmove(&p->h[1], &p->h[0], sizeof(int32_t) * (VLX_NEURAL_TAPS - 1));
    p->h[0] = actual;
}

/* ------------------------------------------------------------ LPC analysis */
/* Autocorrelation (sequential double accumulation, stride 4 above 4096 samples) + Levinson-Durbin,
 * reflection clamp +-0.999, quantise to 2^11 (VeloxCore.h:44-66).  coefs must hold `order` ints.
 * Leaves coefs/shift untouched when n == 0 (:45). */
void vlxo_compute_lpc(const int64_t *x, size_t n, int order, int *coefs, int *shift)
{
    if (n == 0) return;
    double ac[13];
    size_t stride = (n > 4096) ? 4 : 1;
    for (int i = 0; i <= order; i++) {
        double s = 0;
        for (size_t j = (size_t)i; j < n; j += stride) s += (double)x[j] * (double)x[j - i];
        ac[i] = s;
    }
    if (fabs(ac[0]) < 1e-9) { *shift = 0; for (int i = 0; i < order; i++) coefs[i] = 0; return; }
    double a[13][13]; double e[13];
    memset(a, 0, sizeof(a)); memset(e, 0, sizeof(e));
    e[0] = ac[0];
    for (int i = 1; i <= order; i++) {
        double k = ac[i];
        for (int j = 1; j < i; j++) k -= a[j][i - 1] * ac[i - j];
        k /= e[i - 1];
        if (k > 0.999) k = 0.999;
        if (k < -0.999) k = -0.999;
        a[i][i] = k;
        for (int j = 1; j < i; j++) a[j][i] = a[j][i - 1] - k * a[i - j][i - 1];
        e[i] = e[i - 1] * (1 - k * k);
    }
    *shift = 11;
    for (int i = 1; i <= order; i++) coefs[i - 1] = (int)floor(a[i][order] * 2048.0 + 0.5);
}

/* --------------------------------------------------- one stream, both ways */
/* Per-stream analysis results, exposed so kernels can be checked stage by stage. */
typedef struct {
    int32_t silent;
    int32_t shift_lsb;
    int32_t lpc_shift;
    int32_t coefs[VLX_LPC_ORDER];
} vlxo_stream_params;

/* TryCompressChannel (VeloxCore.h:69-127) with IsSilence (VeloxAdvanced.h:14-22) and
 * LSBShifter::Analyze/Apply (VeloxFormat.h:196-219).  Appends to w. */
static void encode_stream(const int64_t *in, size_t n, int high_res, bitw *w, vlxo_stream_params *pp)
{
    int64_t *x = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    uint8_t *low = NULL;
    memcpy(x, in, sizeof(int64_t) * n);
    if (high_res) {                                           /* :73-79 */
        low = (uint8_t *)malloc(n ? n : 1);
        for (size_t i = 0; i < n; i++) { low[i] = (uint8_t)(x[i] & 0xFF); x[i] >>= 8; }
    }
    vlxo_stream_params P; memset(&P, 0, sizeof(P));
    int silent = 1;                                           /* :81 — |s| > 0 anywhere? */
    for (size_t i = 0; i < n; i++) if (x[i] != 0) { silent = 0; break; }
    P.silent = silent;
    if (silent) { bw_write(w, 1, 1); if (pp) *pp = P; free(x); free(low); return; }
    bw_write(w, 0, 1);

    /* LSB shift: trailing zeros of OR|x|, capped at 32 (VeloxFormat.h:196-212) */
    int shift_lsb = 0;
    if (n) {
        uint64_t mask = 0;
        for (size_t i = 0; i < n; i++) { int64_t v = x[i]; mask |= (uint64_t)(v < 0 ? -v : v); }
        if (mask) while ((mask & 1) == 0 && shift_lsb < 32) { mask >>= 1; shift_lsb++; }
    }
    if (shift_lsb > 0) for (size_t i = 0; i < n; i++) x[i] >>= shift_lsb;
    bw_write(w, (uint64_t)shift_lsb, 5);                      /* :86 (32 wraps to 0 in 5 bits) */
    P.shift_lsb = shift_lsb;

    int coefs[VLX_LPC_ORDER]; int lpc_shift = 0;
    for (int j = 0; j < VLX_LPC_ORDER; j++) coefs[j] = 0;
    vlxo_compute_lpc(x, n, VLX_LPC_ORDER, coefs, &lpc_shift); /* :90 */
    bw_write(w, (uint64_t)lpc_shift, 5);
    for (int j = 0; j < VLX_LPC_ORDER; j++) { bw_write(w, (uint64_t)(coefs[j] & 0xFFFF), 16); P.coefs[j] = coefs[j]; }
    P.lpc_shift = lpc_shift;
    if (pp) *pp = P;

    lms12 nn; lms_init(&nn);
    uint64_t run_avg = 512;                                   /* :96 */
    for (size_t i = 0; i < n; i++) {                          /* :98-122 */
        int64_t sum = 0;
        for (int j = 0; j < VLX_LPC_ORDER; j++)
            if (i > (size_t)j) sum += (int64_t)coefs[j] * x[i - 1 - j];
        int32_t pred_lpc = (int32_t)(sum >> lpc_shift);
        int64_t res_lpc = x[i] - pred_lpc;
        int32_t pred_nn = lms_predict(&nn);
        int64_t fin = res_lpc - pred_nn;
        rice_put(w, fin, rice_k(run_avg));
        lms_update(&nn, (int32_t)res_lpc, pred_nn);
        run_avg = run_avg_next(run_avg, zigzag(fin));
    }
    if (high_res) for (size_t i = 0; i < n; i++) bw_write(w, low[i], 8);   /* :124-126 */
    free(x); free(low);
}

/* DecodeChannelWorker (VeloxCore.h:130-172) + LSBShifter::Restore (VeloxFormat.h:220-226). */
static void decode_stream(bitr *r, size_t count, int high_res, int64_t *out)
{
    if (br_bit(r)) { memset(out, 0, sizeof(int64_t) * count); return; }
    int shift_lsb = (int)br_read(r, 5);
    int lpc_shift = (int)br_read(r, 5);
    int coefs[VLX_LPC_ORDER];
    for (int j = 0; j < VLX_LPC_ORDER; j++) coefs[j] = (int)br_read_s(r, 16);
    lms12 nn; lms_init(&nn);
    uint64_t run_avg = 512;
    for (size_t i = 0; i < count; i++) {
        int64_t fin = rice_get(r, rice_k(run_avg));
        int32_t pred_nn = lms_predict(&nn);
        int64_t res_lpc = fin + pred_nn;
        int64_t sum = 0;
        for (int j = 0; j < VLX_LPC_ORDER; j++)
            if (i > (size_t)j) sum += (int64_t)coefs[j] * out[i - 1 - j];
        out[i] = res_lpc + (sum >> lpc_shift);                /* :155 — not truncated to int32 */
        lms_update(&nn, (int32_t)res_lpc, pred_nn);
        run_avg = run_avg_next(run_avg, zigzag(fin));
    }
    if (shift_lsb > 0) for (size_t i = 0; i < count; i++) out[i] = (int64_t)((uint64_t)out[i] << shift_lsb);
    if (high_res)
        for (size_t i = 0; i < count; i++) {
            uint8_t low = (uint8_t)br_read(r, 8);
            out[i] = (int64_t)(((uint64_t)out[i] << 8) | low);
        }
}

size_t vlxo_encode_stream(const int64_t *x, size_t n, int high_res, uint8_t *out, size_t cap,
                          vlxo_stream_params *pp, uint64_t *nbits)
{
    bitw w; bw_init(&w);
    encode_stream(x, n, high_res, &w, pp);
    if (nbits) *nbits = bw_bits(&w);
    bw_flush(&w);
    size_t m = w.len < cap ? w.len : cap;
    if (out) memcpy(out, w.buf, m);
    size_t len = w.len;
    bw_free(&w);
    return len;
}

void vlxo_decode_stream(const uint8_t *d, size_t size, size_t count, int high_res, int64_t *out)
{
    bitr r; br_init(&r, d, size);
    decode_stream(&r, count, high_res, out);
}

/* ------------------------------------------------------------ one sub-block */
/* Sub-block split + M/S decision + chunk body + raw fallback (VeloxCore.h:255-295, :175-177).
 * `inter` holds 2*len interleaved samples.  Produces the flushed chunk bytes in w (fresh). */
static void encode_chunk(const int64_t *inter, size_t len, int high_res, bitw *w, int *use_ms_out)
{
    int64_t *a = (int64_t *)malloc(sizeof(int64_t) * (len ? len : 1));
    int64_t *b = (int64_t *)malloc(sizeof(int64_t) * (len ? len : 1));
    uint64_t sad_lr = 0, sad_ms = 0;
    for (size_t j = 0; j < len; j++) {
        int64_t L = inter[2 * j], R = inter[2 * j + 1];
        a[j] = L; b[j] = R;
        sad_lr += (uint64_t)(L < 0 ? -L : L) + (uint64_t)(R < 0 ? -R : R);
        int64_t mid = (L + R) >> 1, side = L - R;
        sad_ms += (uint64_t)(mid < 0 ? -mid : mid) + (uint64_t)(side < 0 ? -side : side);
    }
    int use_ms = sad_ms < sad_lr;
    if (use_ms) for (size_t j = 0; j < len; j++) { int64_t L = a[j], R = b[j]; a[j] = (L + R) >> 1; b[j] = L - R; }
    if (use_ms_out) *use_ms_out = use_ms;
    bw_write(w, 1, 1); bw_write(w, (uint64_t)use_ms, 1);
    encode_stream(a, len, high_res, w, NULL);
    encode_stream(b, len, high_res, w, NULL);
    bw_flush(w);
    if (w->len >= (len + len) * 5) {                          /* :286-292 raw 40-bit block */
        w->len = 0; w->acc = 0; w->cnt = 0;
        bw_write(w, 0, 1); bw_write(w, (uint64_t)use_ms, 1);
        for (size_t j = 0; j < len; j++) bw_write(w, zigzag(a[j]), 40);
        for (size_t j = 0; j < len; j++) bw_write(w, zigzag(b[j]), 40);
        bw_flush(w);
    }
    free(a); free(b);
}

/* Odd-total path: whole input as one stream, no use_MS bit (VeloxCore.h:242-253). */
static void encode_odd(const int64_t *s, size_t n, int high_res, bitw *w)
{
    bw_write(w, 1, 1);
    encode_stream(s, n, high_res, w, NULL);
    bw_flush(w);
    if (w->len > n * 5) {
        w->len = 0; w->acc = 0; w->cnt = 0;
        bw_write(w, 0, 1);
        for (size_t j = 0; j < n; j++) bw_write(w, zigzag(s[j]), 40);
        bw_flush(w);
    }
}

size_t vlxo_encode_chunk(const int64_t *inter, size_t len, int high_res, uint8_t *out, size_t cap, int *use_ms)
{
    bitw w; bw_init(&w);
    encode_chunk(inter, len, high_res, &w, use_ms);
    size_t m = w.len < cap ? w.len : cap;
    if (out) memcpy(out, w.buf, m);
    size_t l = w.len; bw_free(&w);
    return l;
}

/* --------------------------------------------------------------- float path */
/* DetectPseudoFloat (VeloxFormat.h:14-58): does every probed float round-trip through int16/int24? */
int vlxo_detect_pseudo_float(const uint8_t *raw, size_t count)
{
    int fit16 = 1, fit24 = 1;
    size_t stride = count > 100000 ? 4 : 1;
    for (size_t i = 0; i < count; i += stride) {
        float f; memcpy(&f, raw + 4 * i, 4);
        if (f == 0.0f || !isfinite(f)) continue;
        if (fit16) {
            float s = f * 32768.0f;
            int32_t q = (int32_t)roundf(s);
            float back = (float)q * (1.0f / 32768.0f);
            if (back != f) fit16 = 0;
        }
        if (fit24) {
            float s = f * 8388608.0f;
            int32_t q = (int32_t)roundf(s);
            float back = (float)q * (1.0f / 8388608.0f);
            if (back != f) fit24 = 0;
        }
        if (!fit16 && !fit24) return 0;
    }
    if (fit16) return 16;
    if (fit24) return 24;
    return 0;
}
/* DemoteFloatToInt (VeloxFormat.h:60-70) */
void vlxo_demote_float(const uint8_t *raw, size_t count, int bits, int64_t *out)
{
    double scale = bits == 16 ? 32768.0 : 8388608.0;
    for (size_t i = 0; i < count; i++) { float f; memcpy(&f, raw + 4 * i, 4); out[i] = (int64_t)round((double)f * scale); }
}
/* SplitFloat32 (VeloxFormat.h:84-105) */
void vlxo_split_float32(const uint8_t *raw, size_t count, int64_t *mant, uint8_t *exps)
{
    for (size_t i = 0; i < count; i++) {
        uint32_t u; memcpy(&u, raw + 4 * i, 4);
        uint32_t e = (u >> 23) & 0xFF, m = u & 0x7FFFFF;
        exps[i] = (uint8_t)e;
        if (e) m |= 0x800000;
        mant[i] = (u >> 31) ? -(int64_t)m : (int64_t)m;
    }
}
/* MergeFloat32 (VeloxFormat.h:107-128): what the decoding callers build from (mantissa, exponent) */
void vlxo_merge_float32(const int64_t *mant, const uint8_t *exps, size_t count, uint8_t *out)
{
    for (size_t i = 0; i < count; i++) {
        int64_t m = mant[i]; uint32_t sign = 0;
        if (m < 0) { sign = 1; m = (int64_t)(0ull - (uint64_t)m); }
        uint32_t u = (sign << 31) | ((uint32_t)exps[i] << 23) | (uint32_t)(m & 0x7FFFFF);
        memcpy(out + 4 * i, &u, 4);
    }
}
/* PromoteIntToFloat (VeloxFormat.h:72-82) */
void vlxo_promote_int_to_float(const int64_t *in, size_t count, int bits, uint8_t *out)
{
    float scale = bits == 16 ? (1.0f / 32768.0f) : (1.0f / 8388608.0f);
    for (size_t i = 0; i < count; i++) { float f = (float)in[i] * scale; memcpy(out + 4 * i, &f, 4); }
}
/* EncodeRLE (VeloxCore.h:187-195) */
static void rle_put(const uint8_t *d, size_t n, bitw *w)
{
    if (!n) return;
    uint8_t last = d[0]; int run = 0;
    for (size_t i = 0; i < n; i++) {
        if (d[i] == last && run < 255) run++;
        else { bw_write(w, (uint64_t)run, 8); bw_write(w, last, 8); last = d[i]; run = 1; }
    }
    bw_write(w, (uint64_t)run, 8); bw_write(w, last, 8);
}

/* ------------------------------------------------------------- whole payload */
typedef struct {
    const int64_t *s; size_t total; int high_res;
    size_t nchunks; bitw *chunks;
    size_t next; pthread_mutex_t mu;
} enc_job;

static void *enc_worker(void *arg)
{
    enc_job *J = (enc_job *)arg;
    for (;;) {
        pthread_mutex_lock(&J->mu);
        size_t c = J->next++;
        pthread_mutex_unlock(&J->mu);
        if (c >= J->nchunks) break;
        size_t i = c * VLX_SUB_BLOCK;
        size_t end = i + VLX_SUB_BLOCK < J->total ? i + VLX_SUB_BLOCK : J->total;
        encode_chunk(J->s + i, (end - i) / 2, J->high_res, &J->chunks[c], NULL);
    }
    return NULL;
}

/* Encoder::ProcessBlock (VeloxCore.h:210-306).  `samples` is rewritten for pseudo-float input
 * exactly as the reference does (:219,:222).  Returns malloc'd payload in *out.
 * nthreads <= 1: serial; otherwise a pool over sub-blocks (the reference's ThreadPool, :208). */
size_t vlxo_encode_mt(int64_t *samples, size_t total, int is_float, const uint8_t *exps, size_t n_exps,
                      const uint8_t *raw_bytes, int nthreads, uint8_t **out)
{
    bitw bs; bw_init(&bs);
    int float_mode = 0;
    if (is_float) {                                           /* :215-224 */
        int det = vlxo_detect_pseudo_float(raw_bytes, total);
        if (det == 16) { float_mode = 1; vlxo_demote_float(raw_bytes, total, 16, samples); }
        else if (det == 24) { float_mode = 2; vlxo_demote_float(raw_bytes, total, 24, samples); }
    }
    bw_write(&bs, (uint64_t)(is_float != 0), 1);              /* :226-230 */
    if (is_float) {
        bw_write(&bs, (uint64_t)float_mode, 2);
        if (float_mode == 0) rle_put(exps, n_exps, &bs);
    }
    int high_res = 0;                                         /* :232-236 */
    if (!is_float || float_mode > 0)
        for (size_t i = 0; i < total; i++) { int64_t v = samples[i]; if ((v < 0 ? (int64_t)(0ull - (uint64_t)v) : v) > 65536) { high_res = 1; break; } }   /* std::abs(INT64_MIN) stays negative on x86-64 */
    bw_write(&bs, (uint64_t)high_res, 1);

    if (total % 2 != 0) {                                     /* :242-253 */
        bitw c; bw_init(&c);
        encode_odd(samples, total, high_res, &c);
        bw_write(&bs, (uint32_t)c.len, 32);
        for (size_t k = 0; k < c.len; k++) bw_write(&bs, c.buf[k], 8);
        bw_free(&c);
    } else {
        size_t nchunks = (total + VLX_SUB_BLOCK - 1) / VLX_SUB_BLOCK;
        enc_job J; J.s = samples; J.total = total; J.high_res = high_res; J.nchunks = nchunks; J.next = 0;
        J.chunks = (bitw *)malloc(sizeof(bitw) * (nchunks ? nchunks : 1));
        for (size_t c = 0; c < nchunks; c++) bw_init(&J.chunks[c]);
        pthread_mutex_init(&J.mu, NULL);
        if (nthreads <= 1) enc_worker(&J);
        else {
            pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
            for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, enc_worker, &J);
            for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
            free(th);
        }
        pthread_mutex_destroy(&J.mu);
        for (size_t c = 0; c < nchunks; c++) {                /* :299-303 in submission order */
            bw_write(&bs, (uint32_t)J.chunks[c].len, 32);
            for (size_t k = 0; k < J.chunks[c].len; k++) bw_write(&bs, J.chunks[c].buf[k], 8);
            bw_free(&J.chunks[c]);
        }
        free(J.chunks);
    }
    bw_flush(&bs);
    *out = bs.buf ? bs.buf : (uint8_t *)malloc(1);
    return bs.len;
}

size_t vlxo_encode(int64_t *samples, size_t total, int is_float, const uint8_t *exps, size_t n_exps,
                   const uint8_t *raw_bytes, uint8_t **out)
{ return vlxo_encode_mt(samples, total, is_float, exps, n_exps, raw_bytes, 1, out); }

void vlxo_free(void *p) { free(p); }

/* StreamingDecoder ctor + DecodeNext driven to exhaustion (VeloxCore.h:322-388; main.cpp:253-261).
 * out/out_exp hold `total` entries.  Returns the number of samples produced. */
size_t vlxo_decode(const uint8_t *data, size_t size, size_t total, int64_t *out, uint8_t *out_exp,
                   int *is_float_out, int *float_mode_out)
{
    bitr bs; br_init(&bs, data, size);
    int is_float = (int)br_read(&bs, 1), float_mode = 0;      /* :324-329 */
    uint8_t *exps = NULL; size_t n_exps = 0;
    if (is_float) {
        float_mode = (int)br_read(&bs, 2);
        if (float_mode == 0) {                                /* DecodeRLE :196-203 */
            exps = (uint8_t *)malloc(total + 256);
            while (n_exps < total) {
                int run = (int)br_read(&bs, 8), val = (int)br_read(&bs, 8);
                if (run == 0 && (bs.pos >> 3) >= bs.size) break;   /* reference would spin forever on a truncated file */
                for (int i = 0; i < run; i++) exps[n_exps++] = (uint8_t)val;
            }
        }
    }
    int high_res = (int)br_read(&bs, 1);
    if (is_float_out) *is_float_out = is_float && float_mode == 0;    /* IsFloat() :332 */
    if (float_mode_out) *float_mode_out = float_mode;

    size_t decoded = 0, exp_idx = 0;
    int64_t *c1 = (int64_t *)malloc(sizeof(int64_t) * 4096), *c2 = (int64_t *)malloc(sizeof(int64_t) * 4096);
    while (decoded < total) {
        uint32_t chunk_size = (uint32_t)br_read(&bs, 32);     /* :340-341 */
        if (chunk_size == 0) break;
        /* :343-345 copies chunk bytes out of the (bit-offset) outer stream; bytes past the end read as 0 */
        uint8_t *cd = (uint8_t *)malloc(chunk_size);
        for (uint32_t i = 0; i < chunk_size; i++) cd[i] = (uint8_t)br_read(&bs, 8);
        bitr ch; br_init(&ch, cd, chunk_size);
        int mode = br_bit(&ch);
        size_t remaining = total - decoded;
        size_t frames = remaining / 2 < 4096 ? remaining / 2 : 4096;
        if (frames == 0 && remaining > 0) frames = remaining;
        if (frames > 4096) { c1 = (int64_t *)realloc(c1, sizeof(int64_t) * frames); c2 = (int64_t *)realloc(c2, sizeof(int64_t) * frames); }
        int use_ms = br_bit(&ch);
        if (mode == 1) {
            decode_stream(&ch, frames, high_res, c1);
            decode_stream(&ch, frames, high_res, c2);
        } else {                                              /* ReadRawBlock :179-184 */
            for (size_t j = 0; j < frames; j++) c1[j] = dezigzag(br_read(&ch, 40));
            for (size_t j = 0; j < frames; j++) c2[j] = dezigzag(br_read(&ch, 40));
        }
        for (size_t j = 0; j < frames; j++) {                 /* :357-364 */
            int64_t v0, v1;
            if (use_ms) { v0 = c1[j] + ((c2[j] + 1) >> 1); v1 = c1[j] - (c2[j] >> 1); }
            else { v0 = c1[j]; v1 = c2[j]; }
            int64_t v[2] = { v0, v1 };
            for (int t = 0; t < 2 && decoded < total; t++) {
                out[decoded] = v[t];
                if (out_exp) out_exp[decoded] = (is_float && float_mode == 0 && exp_idx < n_exps) ? exps[exp_idx++] : 0;
                decoded++;
            }
        }
        free(cd);
    }
    free(c1); free(c2); free(exps);
    return decoded;
}

/* ---- chunk-parallel decode for the CPU baseline ("port", all cores) ---------------------- */
typedef struct {
    const uint8_t *data; size_t size; size_t total; int high_res;
    size_t nchunks; const uint64_t *bitpos; const uint32_t *csize; int64_t *out;
    size_t next; pthread_mutex_t mu;
} dec_job;

static void *dec_worker(void *arg)
{
    dec_job *J = (dec_job *)arg;
    int64_t *c1 = (int64_t *)malloc(sizeof(int64_t) * 4096), *c2 = (int64_t *)malloc(sizeof(int64_t) * 4096);
    for (;;) {
        pthread_mutex_lock(&J->mu);
        size_t c = J->next++;
        pthread_mutex_unlock(&J->mu);
        if (c >= J->nchunks) break;
        bitr bs; br_init(&bs, J->data, J->size); bs.pos = J->bitpos[c];
        uint32_t n = J->csize[c];
        uint8_t *cd = (uint8_t *)malloc(n ? n : 1);
        for (uint32_t i = 0; i < n; i++) cd[i] = (uint8_t)br_read(&bs, 8);
        bitr ch; br_init(&ch, cd, n);
        size_t done = c * VLX_SUB_BLOCK, remaining = J->total - done;
        size_t frames = remaining / 2 < 4096 ? remaining / 2 : 4096;
        int mode = br_bit(&ch), use_ms = br_bit(&ch);
        if (mode == 1) { decode_stream(&ch, frames, J->high_res, c1); decode_stream(&ch, frames, J->high_res, c2); }
        else {
            for (size_t j = 0; j < frames; j++) c1[j] = dezigzag(br_read(&ch, 40));
            for (size_t j = 0; j < frames; j++) c2[j] = dezigzag(br_read(&ch, 40));
        }
        for (size_t j = 0; j < frames; j++) {
            if (use_ms) { J->out[done + 2 * j] = c1[j] + ((c2[j] + 1) >> 1); J->out[done + 2 * j + 1] = c1[j] - (c2[j] >> 1); }
            else { J->out[done + 2 * j] = c1[j]; J->out[done + 2 * j + 1] = c2[j]; }
        }
        free(cd);
    }
    free(c1); free(c2);
    return NULL;
}

/* PCM (non-float), even-total payloads only.  Walks the chunk list once, then decodes chunks on
 * nthreads workers.  Returns samples produced (0 if the payload is not of that shape). */
size_t vlxo_decode_mt(const uint8_t *data, size_t size, size_t total, int nthreads, int64_t *out)
{
    bitr bs; br_init(&bs, data, size);
    if (br_read(&bs, 1) != 0 || (total & 1)) return 0;
    int high_res = (int)br_read(&bs, 1);
    size_t nchunks = (total + VLX_SUB_BLOCK - 1) / VLX_SUB_BLOCK;
    uint64_t *bitpos = (uint64_t *)malloc(sizeof(uint64_t) * (nchunks ? nchunks : 1));
    uint32_t *csize = (uint32_t *)malloc(sizeof(uint32_t) * (nchunks ? nchunks : 1));
    size_t found = 0;
    for (; found < nchunks; found++) {
        uint32_t n = (uint32_t)br_read(&bs, 32);
        if (n == 0) break;
        bitpos[found] = bs.pos; csize[found] = n;
        bs.pos += (uint64_t)n * 8;
    }
    dec_job J; J.data = data; J.size = size; J.total = total; J.high_res = high_res; J.nchunks = found;
    J.bitpos = bitpos; J.csize = csize; J.out = out; J.next = 0;
    pthread_mutex_init(&J.mu, NULL);
    if (nthreads <= 1) dec_worker(&J);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, dec_worker, &J);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&J.mu);
    free(bitpos); free(csize);
    size_t produced = found * VLX_SUB_BLOCK;
    return produced < total ? produced : total;
}

/* --------------------------------------------------------- sample (un)packing */
/* BytesToSamples / SamplesToBytes, 16/24/32-bit little-endian (VeloxFormat.h:130-189). */
void vlxo_bytes_to_samples(const uint8_t *b, size_t count, int bits, int64_t *out)
{
    for (size_t i = 0; i < count; i++) {
        if (bits == 16) { int16_t v; memcpy(&v, b + 2 * i, 2); out[i] = v; }
        else if (bits == 24) {
            uint32_t u = (uint32_t)b[3 * i] | ((uint32_t)b[3 * i + 1] << 8) | ((uint32_t)b[3 * i + 2] << 16);
            if (u & 0x800000) u |= 0xFF000000u;
            out[i] = (int32_t)u;
        } else { int32_t v; memcpy(&v, b + 4 * i, 4); out[i] = v; }
    }
}
void vlxo_samples_to_bytes(const int64_t *in, size_t count, int bits, uint8_t *b)
{
    for (size_t i = 0; i < count; i++) {
        int32_t v = (int32_t)in[i];
        if (bits == 16) { int16_t s = (int16_t)in[i]; memcpy(b + 2 * i, &s, 2); }
        else if (bits == 24) { b[3 * i] = v & 0xFF; b[3 * i + 1] = (v >> 8) & 0xFF; b[3 * i + 2] = (v >> 16) & 0xFF; }
        else memcpy(b + 4 * i, &v, 4);
    }
}

/* ------------------------------------------------------ exhaustive payload check */
/* tools/verify_full.py, tests: EVERY chunk of a whole integer-PCM payload against encode_chunk of the same
 * sub-block, on nthreads threads.  The chunk list is walked as the reference's decoder walks it
 * (VeloxCore.h:338-346: u32 size, then that many bytes, all 2 bits off byte alignment behind the two
 * header bits of :226-236); high_res is recomputed over the whole input (:232-236).
 * pcm: packed little-endian samples of `bits` (16 / 24 / 32).  Returns the number of chunks that differ
 * (size prefix or any byte), -1 when the header bits, the chunk count or the total length are wrong;
 * *n_chunks = chunks compared, *first_bad = index of the first differing chunk (or -1). */
typedef struct {
    const uint8_t *pcm; int bits; size_t total; int high_res;
    const uint8_t *payload; size_t size; const size_t *byte_off; size_t nchunks;
    size_t next; size_t bad; long first_bad; pthread_mutex_t mu;
} ver_job;

static void *ver_worker(void *arg)
{
    ver_job *J = (ver_job *)arg;
    const size_t W = (size_t)J->bits / 8;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * VLX_SUB_BLOCK);
    for (;;) {
        pthread_mutex_lock(&J->mu);
        size_t c = J->next++;
        pthread_mutex_unlock(&J->mu);
        if (c >= J->nchunks) break;
        size_t i = c * VLX_SUB_BLOCK;
        size_t end = i + VLX_SUB_BLOCK < J->total ? i + VLX_SUB_BLOCK : J->total;
        vlxo_bytes_to_samples(J->pcm + i * W, end - i, J->bits, s);
        bitw w; bw_init(&w);
        encode_chunk(s, (end - i) / 2, J->high_res, &w, NULL);
        /* the chunk in the payload: [byte_off[c], byte_off[c+1]) shifted left by 2 bits; the first 4 bytes are its size */
        const size_t o = J->byte_off[c], ln = J->byte_off[c + 1] - o;
        int same = ln == w.len + 4;
        if (same) {
            for (size_t k = 0; k < ln && same; k++) {
                const uint8_t got = (uint8_t)((J->payload[o + k] >> 2) | (J->payload[o + k + 1] << 6));
                const uint8_t want = k < 4 ? (uint8_t)((uint32_t)w.len >> (8 * k)) : w.buf[k - 4];
                same = got == want;
            }
        }
        bw_free(&w);
        if (!same) {
            pthread_mutex_lock(&J->mu);
            J->bad++;
            if (J->first_bad < 0 || (long)c < J->first_bad) J->first_bad = (long)c;
            pthread_mutex_unlock(&J->mu);
        }
    }
    free(s);
    return NULL;
}

long vlxo_verify_payload(const uint8_t *pcm, int bits, size_t total, const uint8_t *payload, size_t size,
                         int nthreads, size_t *n_chunks, long *first_bad)
{
    if (n_chunks) *n_chunks = 0;
    if (first_bad) *first_bad = -1;
    if ((total & 1) || !size || (bits != 16 && bits != 24 && bits != 32)) return -1;
    const size_t W = (size_t)bits / 8;
    int high_res = 0;
    if (bits != 16) {
        int64_t v;
        for (size_t i = 0; i < total && !high_res; i++) { vlxo_bytes_to_samples(pcm + i * W, 1, bits, &v); if ((v < 0 ? -v : v) > 65536) high_res = 1; }
    }
    if ((payload[0] & 1) != 0 || ((payload[0] >> 1) & 1) != high_res) return -1;      /* is_float = 0, high_res (:226-236) */
    const size_t nchunks = (total + VLX_SUB_BLOCK - 1) / VLX_SUB_BLOCK;
    size_t *off = (size_t *)malloc(sizeof(size_t) * (nchunks + 1));
    size_t pos = 0;                                            /* byte position of the chunk's size prefix (+ 2 bits) */
    for (size_t c = 0; c < nchunks; c++) {
        off[c] = pos;
        if (pos + 5 > size) { free(off); return -1; }
        uint32_t sz = 0;
        for (int k = 0; k < 4; k++) sz |= (uint32_t)(uint8_t)((payload[pos + k] >> 2) | (payload[pos + k + 1] << 6)) << (8 * k);
        pos += 4 + (size_t)sz;
        if (pos + 1 > size) { free(off); return -1; }
    }
    off[nchunks] = pos;
    if (pos + 1 != size || (payload[pos] >> 2) != 0) { free(off); return -1; }        /* the final Flush (:305): 2 bits + zero padding */
    ver_job J; J.pcm = pcm; J.bits = bits; J.total = total; J.high_res = high_res; J.payload = payload; J.size = size;
    J.byte_off = off; J.nchunks = nchunks; J.next = 0; J.bad = 0; J.first_bad = -1;
    pthread_mutex_init(&J.mu, NULL);
    if (nthreads <= 1) ver_worker(&J);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, ver_worker, &J);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&J.mu);
    free(off);
    if (n_chunks) *n_chunks = nchunks;
    if (first_bad) *first_bad = J.first_bad;
    return (long)J.bad;
}
