// ude_tableaux.h -- Tsit5 / Vern7 coefficient tables as compile-time functions.
// Numbers come from ude_tableaux_gen.h (decoded from the reference's .jld2 artifacts).
// Stage convention: k[0] = f(uprev); k[s] = f(uprev + dt * sum_{j<s} A(s,j) k[j]) for s = 1..S-1;
// u_new = uprev + dt * sum_j B(j) k[j]; err = dt * sum_j BT(j) k[j].
#pragma once
#include "ude_real.h"
#include "ude_tableaux_gen.h"

namespace ude {

#define T_(x) UDE_TSIT5_##x
struct Tsit5Tab {  // Tsit5(): LotkaVolterra/scenario_1.jl:191,202,206; FisherKPP/Fisher-KPP-CNN.jl:66,136
    static constexpr int S = 7, NK = 7, NEXTRA = 0, ORDER = 5;
    static constexpr bool FSAL = true;
    static constexpr double A(int s, int j) {
        constexpr double a[7][7] = {
            {0, 0, 0, 0, 0, 0, 0},
            {T_(a21), 0, 0, 0, 0, 0, 0},
            {T_(a31), T_(a32), 0, 0, 0, 0, 0},
            {T_(a41), T_(a42), T_(a43), 0, 0, 0, 0},
            {T_(a51), T_(a52), T_(a53), T_(a54), 0, 0, 0},
            {T_(a61), T_(a62), T_(a63), T_(a64), T_(a65), 0, 0},
            {T_(a71), T_(a72), T_(a73), T_(a74), T_(a75), T_(a76), 0}};
        return a[s][j];
    }
    static constexpr double C(int s) {
        constexpr double c[7] = {0, T_(c1), T_(c2), T_(c3), T_(c4), T_(c5), T_(c6)};
        return c[s];
    }
    static constexpr double B(int j) { return A(6, j); }
    static constexpr double BT(int j) {
        constexpr double b[7] = {T_(btilde1), T_(btilde2), T_(btilde3), T_(btilde4), T_(btilde5), T_(btilde6), T_(btilde7)};
        return b[j];
    }
    static constexpr double AE(int, int) { return 0; }
    static constexpr double CE(int) { return 0; }
    // dense-output weights b_j(theta), j = 0..6 (free 4th-order interpolant)
    static __device__ __forceinline__ void bth(real th, real* b) {
        // ARITH-SPEC: Horner with fma (upstream @evalpoly uses muladd); coefficients rounded to `real` (the Float32
        // tableaux upstream are the rounded Float64 ones)
#define H3_(p) (th2 * rfma(th, rfma(th, (real)T_(p##4), (real)T_(p##3)), (real)T_(p##2)))
        const real th2 = th * th;
        b[0] = th * rfma(th, rfma(th, rfma(th, (real)T_(r14), (real)T_(r13)), (real)T_(r12)), (real)T_(r11));
        b[1] = H3_(r2);
        b[2] = H3_(r3);
        b[3] = H3_(r4);
        b[4] = H3_(r5);
        b[5] = H3_(r6);
        b[6] = H3_(r7);
#undef H3_
    }
    static constexpr bool dense_uses(int j) { return j < 7; }
    // b_q(theta) as a 7-slot Horner table, highest order first, padded with LEADING zeros (fma(th, 0, c) == c exactly):
    //   h = R(q,0); h = fma(th, h, R(q,i)) i = 1..6;  b_q = (q == 0 ? th : th*th) * h  -- bit-identical to bth()
    static constexpr double R(int q, int i) {
        constexpr double r[7][7] = {
            {0, 0, 0, T_(r14), T_(r13), T_(r12), T_(r11)},
            {0, 0, 0, 0, T_(r24), T_(r23), T_(r22)}, {0, 0, 0, 0, T_(r34), T_(r33), T_(r32)},
            {0, 0, 0, 0, T_(r44), T_(r43), T_(r42)}, {0, 0, 0, 0, T_(r54), T_(r53), T_(r52)},
            {0, 0, 0, 0, T_(r64), T_(r63), T_(r62)}, {0, 0, 0, 0, T_(r74), T_(r73), T_(r72)}};
        return r[q][i];
    }
};
#undef T_

#define V_(x) UDE_VERN7_##x
struct Vern7Tab {  // Vern7(): scenario_1.jl:41,84; SEIR_exposure/seir_exposure.jl:37,138
    static constexpr int S = 10, NK = 16, NEXTRA = 6, ORDER = 7;
    static constexpr bool FSAL = false;
    static constexpr double A(int s, int j) {
        constexpr double a[10][10] = {
            {0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
            {V_(a021), 0, 0, 0, 0, 0, 0, 0, 0, 0},
            {V_(a031), V_(a032), 0, 0, 0, 0, 0, 0, 0, 0},
            {V_(a041), 0, V_(a043), 0, 0, 0, 0, 0, 0, 0},
            {V_(a051), 0, V_(a053), V_(a054), 0, 0, 0, 0, 0, 0},
            {V_(a061), 0, V_(a063), V_(a064), V_(a065), 0, 0, 0, 0, 0},
            {V_(a071), 0, V_(a073), V_(a074), V_(a075), V_(a076), 0, 0, 0, 0},
            {V_(a081), 0, V_(a083), V_(a084), V_(a085), V_(a086), V_(a087), 0, 0, 0},
            {V_(a091), 0, V_(a093), V_(a094), V_(a095), V_(a096), V_(a097), V_(a098), 0, 0},
            {V_(a101), 0, V_(a103), V_(a104), V_(a105), V_(a106), V_(a107), 0, 0, 0}};
        return a[s][j];
    }
    static constexpr double C(int s) {
        constexpr double c[10] = {0, V_(c2), V_(c3), V_(c4), V_(c5), V_(c6), V_(c7), V_(c8), 1.0, 1.0};
        return c[s];
    }
    static constexpr double B(int j) {
        constexpr double b[10] = {V_(b1), 0, 0, V_(b4), V_(b5), V_(b6), V_(b7), V_(b8), V_(b9), 0};
        return b[j];
    }
    static constexpr double BT(int j) {
        constexpr double b[10] = {V_(btilde1), 0, 0, V_(btilde4), V_(btilde5), V_(btilde6), V_(btilde7),
                                  V_(btilde8), V_(btilde9), V_(btilde10)};
        return b[j];
    }
    // lazy dense-output stages k[10..15] (upstream k11..k16): k[10+e] = f(uprev + dt*sum_j AE(e,j) k[j])
    static constexpr double AE(int e, int j) {
        constexpr double a[6][16] = {
            {V_(a1101), 0, 0, V_(a1104), V_(a1105), V_(a1106), V_(a1107), V_(a1108), V_(a1109), 0, 0, 0, 0, 0, 0, 0},
            {V_(a1201), 0, 0, V_(a1204), V_(a1205), V_(a1206), V_(a1207), V_(a1208), V_(a1209), 0, V_(a1211), 0, 0, 0, 0, 0},
            {V_(a1301), 0, 0, V_(a1304), V_(a1305), V_(a1306), V_(a1307), V_(a1308), V_(a1309), 0, V_(a1311), V_(a1312), 0, 0, 0, 0},
            {V_(a1401), 0, 0, V_(a1404), V_(a1405), V_(a1406), V_(a1407), V_(a1408), V_(a1409), 0, V_(a1411), V_(a1412), V_(a1413), 0, 0, 0},
            {V_(a1501), 0, 0, V_(a1504), V_(a1505), V_(a1506), V_(a1507), V_(a1508), V_(a1509), 0, V_(a1511), V_(a1512), V_(a1513), 0, 0, 0},
            {V_(a1601), 0, 0, V_(a1604), V_(a1605), V_(a1606), V_(a1607), V_(a1608), V_(a1609), 0, V_(a1611), V_(a1612), V_(a1613), 0, 0, 0}};
        return a[e][j];
    }
    static constexpr double CE(int e) {
        constexpr double c[6] = {V_(c11), V_(c12), V_(c13), V_(c14), V_(c15), V_(c16)};
        return c[e];
    }
#define F_ rfma
#define W_(x) ((real)V_(x))
#define P6_(p) (th2 * F_(th, F_(th, F_(th, F_(th, F_(th, W_(p##7), W_(p##6)), W_(p##5)), W_(p##4)), W_(p##3)), W_(p##2)))
    static __device__ __forceinline__ void bth(real th, real* b) {
        const real th2 = th * th;
        b[0] = th * F_(th, F_(th, F_(th, F_(th, F_(th, F_(th, W_(r017), W_(r016)), W_(r015)), W_(r014)), W_(r013)), W_(r012)), W_(r011));
        b[1] = 0; b[2] = 0; b[9] = 0;
        b[3] = P6_(r04); b[4] = P6_(r05); b[5] = P6_(r06); b[6] = P6_(r07); b[7] = P6_(r08); b[8] = P6_(r09);
        b[10] = P6_(r11); b[11] = P6_(r12); b[12] = P6_(r13); b[13] = P6_(r14); b[14] = P6_(r15); b[15] = P6_(r16);
    }
#undef W_
#undef P6_
#undef F_
    static constexpr bool dense_uses(int j) { return !(j == 1 || j == 2 || j == 9); }
    // 7-slot Horner table of b_q(theta) (see Tsit5Tab::R)
    static constexpr double R(int q, int i) {
        constexpr double r[16][7] = {
            {V_(r017), V_(r016), V_(r015), V_(r014), V_(r013), V_(r012), V_(r011)},
            {0, 0, 0, 0, 0, 0, 0},
            {0, 0, 0, 0, 0, 0, 0},
            {0, V_(r047), V_(r046), V_(r045), V_(r044), V_(r043), V_(r042)},
            {0, V_(r057), V_(r056), V_(r055), V_(r054), V_(r053), V_(r052)},
            {0, V_(r067), V_(r066), V_(r065), V_(r064), V_(r063), V_(r062)},
            {0, V_(r077), V_(r076), V_(r075), V_(r074), V_(r073), V_(r072)},
            {0, V_(r087), V_(r086), V_(r085), V_(r084), V_(r083), V_(r082)},
            {0, V_(r097), V_(r096), V_(r095), V_(r094), V_(r093), V_(r092)},
            {0, 0, 0, 0, 0, 0, 0},
            {0, V_(r117), V_(r116), V_(r115), V_(r114), V_(r113), V_(r112)},
            {0, V_(r127), V_(r126), V_(r125), V_(r124), V_(r123), V_(r122)},
            {0, V_(r137), V_(r136), V_(r135), V_(r134), V_(r133), V_(r132)},
            {0, V_(r147), V_(r146), V_(r145), V_(r144), V_(r143), V_(r142)},
            {0, V_(r157), V_(r156), V_(r155), V_(r154), V_(r153), V_(r152)},
            {0, V_(r167), V_(r166), V_(r165), V_(r164), V_(r163), V_(r162)}};
        return r[q][i];
    }
};
#undef V_

}  // namespace ude
