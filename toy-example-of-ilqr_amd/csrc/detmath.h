/*
 * detmath.h — deterministic FP64 elementary functions for the CILQR hot path.
 *
 * Why this exists: the CILQR solve loop (reference src/cilqr_solver.cpp:85-153) takes discrete
 * decisions (line-search accept, convergence, PD test, nearest-sample scan) on quantities that
 * differ at the 1e-12 level, so "matches the CPU solver to 1e-5" is only robust when host and
 * device evaluate exp/sin/cos/tan/atan/hypot to the SAME BITS.  libm (glibc) and the ROCm device
 * library (OCML) are each < 1 ulp but not bit-identical to each other.  This header is one
 * implementation, compiled by gcc for the CPU oracle ("detmath" build) and by hipcc for gfx950:
 *
 *   - only IEEE-754 binary64 +,-,*,/,sqrt and EXPLICIT fma() are used;
 *   - both compilers must be run with -ffp-contract=off (no implicit fusing) and no fast-math;
 *   - no tables, no data-dependent branches in the fast paths (selects only) — wave-friendly.
 *
 * Accuracy (checked in tests/test_detmath.py against libm): exp, sin, cos, atan <= 2 ulp,
 * tan <= 4 ulp on the solver's working ranges.  The trigonometric functions are accurate for
 * |x| <~ 2^30; beyond that they return meaningless but still host/device-identical values, and
 * NaN for non-finite input.
 *
 * The polynomial kernels use the classic fdlibm minimax coefficient sets (public constants).
 */
#ifndef CILQR_DETMATH_H
#define CILQR_DETMATH_H

#if defined(__HIPCC__)
#define DM_FN __host__ __device__ static inline
#else
#define DM_FN static inline
#endif

#define DM_FMA(a, b, c) __builtin_fma((a), (b), (c))

/* Device flavours of the trigonometric functions, selected by a template argument that only the HIP
 * compilation sees (for the C oracle build everything below expands to the plain function):
 *   bit 0 (DM_PIN): coefficients pinned to vector registers instead of scalar-register literals.  Inside
 *       the rollout loop the literals (two SGPRs each, ~40 in dm_sincos + dm_tan) overflow the scalar file
 *       and are re-materialised and spilled every step; pinned, they are loaded once before the loop.
 *   bit 1 (DM_NOSHORT): no wave-uniform small-argument shortcut (see DM_WAVE_ALL below) — for loops that
 *       have already made that decision themselves and want straight-line code.
 *   bit 2 (DM_SMALL): the caller guarantees the shortcut's precondition on every active lane (|x| < 0.785
 *       for sin/cos/tan, |x| < 7/16 for atan): take it without asking.
 * Same operations, same bits, in every flavour. */
#define DM_PIN 1
#define DM_NOSHORT 2
#define DM_SMALL 4
#if defined(__HIPCC__)
#define DM_TFN template <int VK = 0> DM_FN
#define DM_T(fn) fn<VK>
/* The coefficients of the DM_PIN flavour: made opaque ONCE (dm_pin_load, before the loop that uses them) and then
 * read in place — an opaque value per use would cost a register copy per use. */
struct DmPinned {
    double S1, S2, S3, S4, S5, S6, C1, C2, C3, C4, C5, C6;   /* dm_ksincos */
    double INV_PIO2, P1, P2, P3, P4, MAGIC;                  /* dm_trig_reduce */
};
static __device__ inline void dm_pin_load(DmPinned& k) {
    k.S1 = -1.66666666666666324348e-01; k.S2 = 8.33333333332248946124e-03; k.S3 = -1.98412698298579493134e-04;
    k.S4 = 2.75573137070700676789e-06; k.S5 = -2.50507602534068634195e-08; k.S6 = 1.58969099521155010221e-10;
    k.C1 = 4.16666666666666019037e-02; k.C2 = -1.38888888888741095749e-03; k.C3 = 2.48015872894767294178e-05;
    k.C4 = -2.75573143513906633035e-07; k.C5 = 2.08757232129817482790e-09; k.C6 = -1.13596475577881948265e-11;
    k.INV_PIO2 = 6.36619772367581382433e-01; k.P1 = 1.57079632673412561417e+00; k.P2 = 6.07710050630396597660e-11;
    k.P3 = 2.02226624871116645580e-21; k.P4 = 8.47842766036889956997e-32; k.MAGIC = 6755399441055744.0;
#if defined(__HIP_DEVICE_COMPILE__)
    /* the same literals as in the functions below (tests compare the pinned flavour with the plain one bit for
     * bit); from here on the compiler sees 18 values in vector registers, not constants */
    __asm__("" : "+v"(k.S1), "+v"(k.S2), "+v"(k.S3), "+v"(k.S4), "+v"(k.S5), "+v"(k.S6));
    __asm__("" : "+v"(k.C1), "+v"(k.C2), "+v"(k.C3), "+v"(k.C4), "+v"(k.C5), "+v"(k.C6));
    __asm__("" : "+v"(k.INV_PIO2), "+v"(k.P1), "+v"(k.P2), "+v"(k.P3), "+v"(k.P4), "+v"(k.MAGIC));
#endif
}
#define DM_PKARG , const DmPinned* pk = nullptr
#define DM_PK , pk
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_K(name, x) ((VK & DM_PIN) ? pk->name : (x))
/* fma(a, b, k) with a polynomial coefficient k as the addend.  Left to the compiler this becomes the two-address
 * v_fmac_f64, which accumulates into the addend's register: a pinned coefficient is first copied (one extra vector
 * instruction per Horner step), a literal one is first built in vector registers (two extra).  The three-operand
 * v_fma_f64 reads the coefficient in place — from its vector register (pinned flavour) or from a scalar register
 * pair, which the scalar unit fills without costing a vector issue slot.  Same operation, same bits. */
static __device__ inline double dm_fma_vk(double a, double b, double k) {
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(k));
    return r;
}
static __device__ inline double dm_fma_sk(double a, double b, double k) {
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
template <int VK>
static __device__ inline double dm_fmak(double a, double b, double k) {
    return (VK & DM_PIN) ? dm_fma_vk(a, b, k) : dm_fma_sk(a, b, k);
}
#define DM_FMAK(a, b, k) dm_fmak<VK>((a), (b), (k))
#define DM_FMAC(a, b, k) dm_fma_sk((a), (b), (k)) /* literal coefficient */
#else
#define DM_K(name, x) (x)
#define DM_FMAK(a, b, k) DM_FMA((a), (b), (k))
#define DM_FMAC(a, b, k) DM_FMA((a), (b), (k))
#endif
#else
#define DM_TFN DM_FN
#define DM_T(fn) fn
#define DM_PKARG
#define DM_PK
#define DM_K(name, x) (x)
#define DM_FMAK(a, b, k) DM_FMA((a), (b), (k))
#define DM_FMAC(a, b, k) DM_FMA((a), (b), (k))
#endif

/* Wave-uniform shortcuts (device only).  A shortcut is taken when EVERY active lane qualifies, and it
 * returns exactly what the general path returns for such arguments — the general path's selects and
 * range reduction degenerate to the identity there — so host and device still agree bit for bit.
 * The host always takes the general path. */
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_WAVE_ALL(cond) (__builtin_amdgcn_ballot_w64(!(cond)) == 0ULL)
#else
#define DM_WAVE_ALL(cond) (cond) /* only so that __device__ code parses in hipcc's host pass */
#endif

DM_FN double dm_from_bits(unsigned long long u) {
    double d;
    __builtin_memcpy(&d, &u, sizeof(d));
    return d;
}

DM_FN unsigned long long dm_to_bits(double d) {
    unsigned long long u;
    __builtin_memcpy(&u, &d, sizeof(u));
    return u;
}

DM_FN double dm_nan(void) { return dm_from_bits(0x7ff8000000000000ULL); }
DM_FN double dm_inf(void) { return dm_from_bits(0x7ff0000000000000ULL); }

/* 2^k for k in [-1022, 1023] */
DM_FN double dm_pow2i(int k) { return dm_from_bits((unsigned long long)(k + 1023) << 52); }

/* double -> int32 with the GPU's conversion semantics on both sides: saturating, NaN -> 0
 * (v_cvt_i32_f64 does this natively; x86's cvttsd2si would return INT_MIN instead) */
DM_FN int dm_f2i_sat(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)x;
#else
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (-2147483647 - 1);
    return (int)x;
#endif
}

/* p * 2^k, k in [-1100, 1100], |p| in [0.5, 2]: one correctly rounded scaling (v_ldexp_f64 on the
 * device; on the host two multiplications, the first of which is exact) */
DM_FN double dm_scale2(double p, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ldexp(p, k);
#else
    int k1 = k >> 1;
    int k2 = k - k1;
    return (p * dm_pow2i(k1)) * dm_pow2i(k2);
#endif
}

DM_FN double dm_sqrt(double x) { return __builtin_sqrt(x); }

/* reference call sites use hypot() from libm (src/cilqr_solver.cpp:237,299,509,528);
 * magnitudes here are metres, so the naive form neither overflows nor underflows. */
DM_FN double dm_hypot(double x, double y) { return dm_sqrt(x * x + y * y); }

DM_FN double dm_exp(double x) {
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2HI = 6.93147180369123816490e-01; /* 32 trailing zero bits */
    const double LN2LO = 1.90821492927058770002e-10;
    const double MAGIC = 6755399441055744.0; /* 1.5 * 2^52: round-to-nearest-even integer */
    /* clamp so that the scaling exponent stays in range: exp(720) overflows to +inf and exp(-760)
     * underflows to +0 through the ordinary arithmetic below; NaN passes through the compares */
    double xc = x;
    xc = (xc > 720.0) ? 720.0 : xc;
    xc = (xc < -760.0) ? -760.0 : xc;
    double kd = (xc * LOG2E + MAGIC) - MAGIC;
    double r = DM_FMA(-kd, LN2HI, xc);
    r = DM_FMA(-kd, LN2LO, r);
    /* exp(r), |r| <= 0.3466: Taylor to r^13 (remainder < 5e-18 relative) */
    double p = 1.6059043836821613e-10;            /* 1/13! */
    p = DM_FMAC(p, r, 2.08767569878681e-09);       /* 1/12! */
    p = DM_FMAC(p, r, 2.505210838544172e-08);      /* 1/11! */
    p = DM_FMAC(p, r, 2.755731922398589e-07);      /* 1/10! */
    p = DM_FMAC(p, r, 2.7557319223985893e-06);     /* 1/9!  */
    p = DM_FMAC(p, r, 2.48015873015873e-05);       /* 1/8!  */
    p = DM_FMAC(p, r, 1.984126984126984e-04);      /* 1/7!  */
    p = DM_FMAC(p, r, 1.388888888888889e-03);      /* 1/6!  */
    p = DM_FMAC(p, r, 8.333333333333333e-03);      /* 1/5!  */
    p = DM_FMAC(p, r, 4.1666666666666664e-02);     /* 1/4!  */
    p = DM_FMAC(p, r, 1.6666666666666666e-01);     /* 1/3!  */
    p = DM_FMA(p, r, 0.5);
    p = DM_FMA(p, r, 1.0);
    p = DM_FMA(p, r, 1.0);
    return dm_scale2(p, dm_f2i_sat(kd)); /* NaN in -> NaN out (p is NaN) */
}

/* ---- trigonometric range reduction: x = n*(pi/2) + r, |r| <= pi/4 (+eps) ---- */
DM_TFN double dm_trig_reduce(double x, int* quadrant DM_PKARG) {
    const double INV_PIO2 = DM_K(INV_PIO2, 6.36619772367581382433e-01);
    const double P1 = DM_K(P1, 1.57079632673412561417e+00);  /* first 33 bits of pi/2 */
    const double P2 = DM_K(P2, 6.07710050630396597660e-11);  /* next 33 bits */
    const double P3 = DM_K(P3, 2.02226624871116645580e-21);  /* next 33 bits */
    const double P4 = DM_K(P4, 8.47842766036889956997e-32);  /* remainder */
    const double MAGIC = DM_K(MAGIC, 6755399441055744.0);
    /* accurate for |x| <~ 2^30; beyond that the result is meaningless but still the same on host
     * and device (saturating quadrant conversion), and +-inf / NaN give NaN through r */
    double nd = (x * INV_PIO2 + MAGIC) - MAGIC;
    double r = DM_FMA(-nd, P1, x);
    r = DM_FMA(-nd, P2, r);
    r = DM_FMA(-nd, P3, r);
    r = DM_FMA(-nd, P4, r);
    *quadrant = dm_f2i_sat(nd) & 3;
    return r;
}

/* sin and cos kernels on |r| <= pi/4 share z = r*r */
DM_TFN void dm_ksincos(double r, double* s_out, double* c_out DM_PKARG) {
    const double S1 = DM_K(S1, -1.66666666666666324348e-01);
    const double S2 = DM_K(S2, 8.33333333332248946124e-03);
    const double S3 = DM_K(S3, -1.98412698298579493134e-04);
    const double S4 = DM_K(S4, 2.75573137070700676789e-06);
    const double S5 = DM_K(S5, -2.50507602534068634195e-08);
    const double S6 = DM_K(S6, 1.58969099521155010221e-10);
    const double C1 = DM_K(C1, 4.16666666666666019037e-02);
    const double C2 = DM_K(C2, -1.38888888888741095749e-03);
    const double C3 = DM_K(C3, 2.48015872894767294178e-05);
    const double C4 = DM_K(C4, -2.75573143513906633035e-07);
    const double C5 = DM_K(C5, 2.08757232129817482790e-09);
    const double C6 = DM_K(C6, -1.13596475577881948265e-11);
    double z = r * r;
    double p = DM_FMAK(z, S6, S5);
    p = DM_FMAK(z, p, S4);
    p = DM_FMAK(z, p, S3);
    p = DM_FMAK(z, p, S2);
    p = DM_FMAK(z, p, S1);
    double v = z * r;
    *s_out = DM_FMA(v, p, r);
    double q = DM_FMAK(z, C6, C5);
    q = DM_FMAK(z, q, C4);
    q = DM_FMAK(z, q, C3);
    q = DM_FMAK(z, q, C2);
    q = DM_FMAK(z, q, C1);
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    double t = z * q;
    *c_out = w + (((1.0 - w) - hz) + z * t);
}

/* flip the sign of d when bit is non-zero (bit is 0 or any value with the wanted truth) */
DM_FN double dm_negate_if(double d, int cond) {
    return dm_from_bits(dm_to_bits(d) ^ ((unsigned long long)(cond != 0) << 63));
}

DM_TFN void dm_sincos(double x, double* s_out, double* c_out DM_PKARG) {
#if defined(__HIP_DEVICE_COMPILE__)
    /* |x| < 0.785 < pi/4 on every lane: x * (2/pi) < 0.49975, so nd = (v + MAGIC) - MAGIC = 0, r = fma(-0, P, x)
     * = x (also for x = +-0), quadrant 0, and the quadrant selection / sign flips below are the identity */
    if ((VK & DM_SMALL) || (!(VK & DM_NOSHORT) && DM_WAVE_ALL(__builtin_fabs(x) < 0.785))) {
        DM_T(dm_ksincos)(x, s_out, c_out DM_PK);
        return;
    }
#endif
    int q;
    double r = DM_T(dm_trig_reduce)(x, &q DM_PK);
    double s, c;
    DM_T(dm_ksincos)(r, &s, &c DM_PK);
    double ss = (q & 1) ? c : s;
    double cc = (q & 1) ? s : c;
    *s_out = dm_negate_if(ss, q & 2);
    *c_out = dm_negate_if(cc, (q + 1) & 2);
}

DM_TFN double dm_sin(double x DM_PKARG) {
    double s, c;
    DM_T(dm_sincos)(x, &s, &c DM_PK);
    return s;
}

DM_TFN double dm_cos(double x DM_PKARG) {
    double s, c;
    DM_T(dm_sincos)(x, &s, &c DM_PK);
    return c;
}

DM_TFN double dm_tan(double x DM_PKARG) {
#if defined(__HIP_DEVICE_COMPILE__)
    if ((VK & DM_SMALL) || (!(VK & DM_NOSHORT) && DM_WAVE_ALL(__builtin_fabs(x) < 0.785))) { /* as in dm_sincos: r = x, quadrant 0, tan = sin / cos */
        double s0, c0;
        DM_T(dm_ksincos)(x, &s0, &c0 DM_PK);
        return s0 / c0;
    }
#endif
    int q;
    double r = DM_T(dm_trig_reduce)(x, &q DM_PK);
    double s, c;
    DM_T(dm_ksincos)(r, &s, &c DM_PK);
    double num = (q & 1) ? c : s;
    double den = (q & 1) ? s : c;
    return dm_negate_if(num, q & 1) / den;
}

DM_TFN double dm_atan(double x DM_PKARG) {
    const double aT0 = 3.33333333333329318027e-01;
    const double aT1 = -1.99999999998764832476e-01;
    const double aT2 = 1.42857142725034663711e-01;
    const double aT3 = -1.11111104054623557880e-01;
    const double aT4 = 9.09088713343650656196e-02;
    const double aT5 = -7.69187620504482999495e-02;
    const double aT6 = 6.66107313738753120669e-02;
    const double aT7 = -5.83357013379057348645e-02;
    const double aT8 = 4.97687799461593236017e-02;
    const double aT9 = -3.65315727442169155270e-02;
    const double aT10 = 1.62858201153657823623e-02;
    double ax = (x < 0.0) ? -x : x;
    int id = -1;
    double hi = 0.0, lo = 0.0, t;
#if defined(__HIP_DEVICE_COMPILE__)
    /* every lane below 7/16: the general path selects id = -1, num = ax, den = 1, t = ax / 1 = ax */
    if ((VK & DM_SMALL) || (!(VK & DM_NOSHORT) && DM_WAVE_ALL(ax < 0.4375))) {
        t = ax;
    } else
#endif
    {
    /* interval selection (fdlibm s_atan.c): 7/16, 11/16, 19/16, 39/16 */
    id = (ax >= 0.4375) ? 0 : id;
    id = (ax >= 0.6875) ? 1 : id;
    id = (ax >= 1.1875) ? 2 : id;
    id = (ax >= 2.4375) ? 3 : id;
    double num = ax, den = 1.0;
    num = (id == 0) ? (2.0 * ax - 1.0) : num;
    den = (id == 0) ? (2.0 + ax) : den;
    hi = (id == 0) ? 4.63647609000806093515e-01 : hi;
    lo = (id == 0) ? 2.26987774529616870924e-17 : lo;
    num = (id == 1) ? (ax - 1.0) : num;
    den = (id == 1) ? (ax + 1.0) : den;
    hi = (id == 1) ? 7.85398163397448278999e-01 : hi;
    lo = (id == 1) ? 3.06161699786838301793e-17 : lo;
    num = (id == 2) ? (ax - 1.5) : num;
    den = (id == 2) ? (1.0 + 1.5 * ax) : den;
    hi = (id == 2) ? 9.82793723247329054082e-01 : hi;
    lo = (id == 2) ? 1.39033110312309984516e-17 : lo;
    num = (id == 3) ? -1.0 : num;
    den = (id == 3) ? ax : den;
    hi = (id == 3) ? 1.57079632679489655800e+00 : hi;
    lo = (id == 3) ? 6.12323399573676603587e-17 : lo;
    t = num / den; /* for id == -1 this is ax / 1.0 == ax exactly */
    }
    double z = t * t;
    double w = z * z;
    double s1 = DM_FMAC(w, aT10, aT8);
    s1 = DM_FMAC(w, s1, aT6);
    s1 = DM_FMAC(w, s1, aT4);
    s1 = DM_FMAC(w, s1, aT2);
    s1 = DM_FMAC(w, s1, aT0);
    s1 = z * s1;
    double s2 = DM_FMAC(w, aT9, aT7);
    s2 = DM_FMAC(w, s2, aT5);
    s2 = DM_FMAC(w, s2, aT3);
    s2 = DM_FMAC(w, s2, aT1);
    s2 = w * s2;
    double corr = t * (s1 + s2);
    double res = (id < 0) ? (t - corr) : (hi - ((corr - lo) - t));
    res = (ax > 1.0e300) ? 1.57079632679489655800e+00 : res; /* incl. +-inf: 1/inf -> 0 handled too */
    res = (x < 0.0) ? -res : res;
    return (x != x) ? x : res;
}

#endif /* CILQR_DETMATH_H */
