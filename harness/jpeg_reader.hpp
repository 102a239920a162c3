// JPEG reader for the harness (the reference loads its images with cv::imread, src/main.cpp:93,158; every bundled dataset image is a JPEG,
// nine of the twelve progressive).  cv::imread decodes with libjpeg(-turbo) at its defaults, and those defaults are all INTEGER
// algorithms with published definitions, so this decoder restates them and is checked bit for bit against Pillow's libjpeg-turbo on
// the twelve dataset files and on synthesized streams (tests/test_harness_jpeg.py):
//   * Huffman entropy decoding, sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection + successive approximation),
//     interleaved and single-component scans, restart intervals, 8- and 16-bit quantisation tables       (ITU-T T.81 annexes F, G)
//   * the "slow integer" inverse DCT: 13-bit constants, 2 extra bits between the passes                   (Loeffler-Ligtenberg-Moschytz;
//     libjpeg's default JDCT_ISLOW) with its wrap-around range limit
//   * "fancy" chroma upsampling: the 3/4 : 1/4 triangle filter, h2v1 / h2v2 / h1v2, edge rows and columns replicated, the rounding
//     constants alternating 8 / 7 (1 / 2) so that the bias averages out; pixel replication for every other ratio
//   * YCbCr -> RGB with 16-bit fixed-point constants (1.402, 0.34414, 0.71414, 1.772), the two chroma terms of G summed before the shift
// Not read: arithmetic coding, lossless and hierarchical processes, 12-bit samples, CMYK / YCCK (refused, never guessed at).
// Every length and index comes from the file: each is checked before use (tests run this under ASan / UBSan on damaged files).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rtdd_jpeg {

static const unsigned char kNatural[64 + 16] = {                    // zigzag position -> natural (row-major) position; 16 spare entries catch a run past 63
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    bool defined = false;
    unsigned char vals[256];
    int maxcode[18], valptr[17];                                    // per code length 1..16 (maxcode[17] = sentinel)
    short look[512];                                                // 9-bit lookahead: (length << 8) | value, or -1
    bool build(const unsigned char counts[16], const unsigned char *symbols, int nsym) {
        int total = 0;
        for (int i = 0; i < 16; i++) total += counts[i];
        if (total != nsym || total > 256) return false;
        std::memcpy(vals, symbols, (size_t)nsym);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k - code;
            if (counts[l - 1]) {
                if (code + counts[l - 1] > (1 << l)) return false;  // over-subscribed
                k += counts[l - 1]; code += counts[l - 1];
                maxcode[l] = code - 1;
            } else maxcode[l] = -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        for (int i = 0; i < 512; i++) look[i] = -1;
        code = 0; k = 0;
        for (int l = 1; l <= 9; l++) {
            for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
                const int first = code << (9 - l);
                for (int j = 0; j < (1 << (9 - l)); j++) look[first + j] = (short)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
        defined = true;
        return true;
    }
};

struct BitReader {
    const unsigned char *p = nullptr, *end = nullptr;
    uint32_t acc = 0; int nbits = 0;
    bool hit_marker = false;                                        // entropy data ran into a marker (or the file's end): zeros from here on
    void start(const unsigned char *b, const unsigned char *e) { p = b; end = e; acc = 0; nbits = 0; hit_marker = false; }
    void fill() {
        while (nbits <= 24) {
            unsigned c = 0;
            if (!hit_marker) {
                if (p >= end) hit_marker = true;
                else if (*p != 0xFF) c = *p++;
                else if (p + 1 < end && p[1] == 0x00) { c = 0xFF; p += 2; }
                else hit_marker = true;                             // a marker: left in place for the segment parser
            }
            acc |= c << (24 - nbits); nbits += 8;
        }
    }
    int peek(int n) { if (nbits < n) fill(); return (int)(acc >> (32 - n)); }
    void drop(int n) { acc <<= n; nbits -= n; }
    int bits(int n) { if (!n) return 0; const int v = peek(n); drop(n); return v; }
    int bit() { return bits(1); }
    int decode(const Huffman &h) {
        const int idx = peek(9);
        const short e = h.look[idx];
        if (e >= 0) { drop(e >> 8); return e & 255; }
        int code = peek(16), l = 10;
        for (; l <= 16; l++) if ((code >> (16 - l)) <= h.maxcode[l]) break;
        if (l > 16) { drop(16); return 0; }                         // no such code: libjpeg warns and uses 0
        drop(l);
        const int at = h.valptr[l] + (code >> (16 - l));
        return (at >= 0 && at < 256) ? h.vals[at] : 0;
    }
    void align() { acc = 0; nbits = 0; }
};

static inline int extend(int x, int s) { return s && x < (1 << (s - 1)) ? x - (1 << s) + 1 : x; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int dw = 0, dh = 0;                                             // samples this component really has (ceil(W h / hmax), ...)
    int bw = 0, bh = 0;                                             // blocks of a single-component scan
    int pw = 0, ph = 0;                                             // blocks stored (a whole number of MCUs)
    std::vector<short> coef;                                        // ph x pw x 64, natural order
    uint16_t quant[64]; bool latched = false;
    int dc_tbl = 0, ac_tbl = 0, pred = 0;
    std::vector<unsigned char> plane;                               // (8 ph) x (8 pw) samples after the inverse DCT
};

struct Decoder {
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false, jfif = false, adobe = false; int adobe_transform = 0;
    uint16_t qt[4][64]; bool qt_defined[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    Component comp[3];
    int restart_interval = 0;
    size_t file_size = 0;
    std::string error;

    bool fail(const char *why) { error = why; return false; }

    // ---- the slow-but-accurate integer inverse DCT (constants 13 bits, 2 extra bits kept between the passes) ----
    // (64-bit intermediates: the same values as libjpeg's 32-bit ones on every stream an encoder can produce, and no overflow on the others)
    typedef long long i64;
    static inline i64 descale(i64 x, int n) { return (x + ((i64)1 << (n - 1))) >> n; }
    static inline unsigned char range_limit(i64 v) {                // libjpeg's table: index (x & 1023) around a centre of 128
        const int x = (int)(v & 1023);
        if (x < 128) return (unsigned char)(x + 128);
        if (x < 512) return 255;
        if (x < 896) return 0;
        return (unsigned char)(x - 896);
    }
    // one 8-point pass: even part from s0 s2 s4 s6, odd part from s1 s3 s5 s7; o[k] = even[k] + odd[k], o[7 - k] = even[k] - odd[k] (unscaled)
    static inline void idct8(const i64 s[8], i64 o[8]) {
        constexpr i64 F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                      F2053 = 16819, F2562 = 20995, F3072 = 25172;
        i64 z1 = (s[2] + s[6]) * F0541;
        const i64 tmp2 = z1 - s[6] * F1847, tmp3 = z1 + s[2] * F0765;
        const i64 tmp0 = (s[0] + s[4]) * 8192, tmp1 = (s[0] - s[4]) * 8192;
        const i64 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        i64 t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
        z1 = t0 + t3;
        i64 z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
        const i64 z5 = (z3 + z4) * F1175;
        t0 *= F0298; t1 *= F2053; t2 *= F3072; t3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        o[0] = tmp10 + t3; o[7] = tmp10 - t3; o[1] = tmp11 + t2; o[6] = tmp11 - t2;
        o[2] = tmp12 + t1; o[5] = tmp12 - t1; o[3] = tmp13 + t0; o[4] = tmp13 - t0;
    }
    static void idct(const short *in, const uint16_t *q, unsigned char *out, int stride) {
        i64 ws[64], s[8], o[8];
        for (int c = 0; c < 8; c++) {                               // columns, dequantising; 2 extra bits kept
            for (int k = 0; k < 8; k++) s[k] = (i64)in[8 * k + c] * q[8 * k + c];
            idct8(s, o);
            for (int k = 0; k < 8; k++) ws[8 * k + c] = descale(o[k], 11);
        }
        for (int r = 0; r < 8; r++) {                               // rows; 13 + 2 + 3 bits out
            idct8(ws + 8 * r, o);
            for (int k = 0; k < 8; k++) out[(size_t)r * stride + k] = range_limit(descale(o[k], 18));
        }
    }

    // ---- segments ----
    bool read_dqt(const unsigned char *d, int len) {
        while (len > 0) {
            const int pq = d[0] >> 4, tq = d[0] & 15, need = 1 + 64 * (pq ? 2 : 1);
            if (pq > 1 || tq > 3 || len < need) return fail("bad DQT");
            for (int i = 0; i < 64; i++) qt[tq][kNatural[i]] = pq ? (uint16_t)((d[1 + 2 * i] << 8) | d[2 + 2 * i]) : d[1 + i];
            qt_defined[tq] = true;
            d += need; len -= need;
        }
        return true;
    }
    bool read_dht(const unsigned char *d, int len) {
        while (len > 0) {
            if (len < 17) return fail("bad DHT");
            const int tc = d[0] >> 4, th = d[0] & 15;
            int n = 0;
            for (int i = 0; i < 16; i++) n += d[1 + i];
            if (tc > 1 || th > 3 || n > 256 || len < 17 + n) return fail("bad DHT");
            if (!(tc ? ac : dc)[th].build(d + 1, d + 17, n)) return fail("bad Huffman table");
            d += 17 + n; len -= 17 + n;
        }
        return true;
    }
    bool read_sof(const unsigned char *d, int len, bool prog) {
        if (have_frame) return fail("second frame header");
        if (len < 6) return fail("bad SOF");
        if (d[0] != 8) return fail("only 8-bit samples are read");
        H = (d[1] << 8) | d[2]; W = (d[3] << 8) | d[4]; ncomp = d[5];
        if (W <= 0 || H <= 0) return fail("empty image (or a DNL-defined height)");
        // (nothing is allocated for a header the file cannot back: the DC scan alone costs a bit per block, i.e. a byte per 512 pixels)
        if ((unsigned long long)W * H > (1ull << 28) || (unsigned long long)W * H / 4096ull > (unsigned long long)file_size + 1024ull) return fail("header promises more than the file can hold");
        if (ncomp != 1 && ncomp != 3) return fail("only gray and three-component images are read");
        if (len < 6 + 3 * ncomp) return fail("bad SOF");
        hmax = vmax = 1;
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.id = d[6 + 3 * i]; c.h = d[7 + 3 * i] >> 4; c.v = d[7 + 3 * i] & 15; c.tq = d[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad sampling factors");
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            if (hmax % c.h || vmax % c.v) return fail("fractional sampling ratios are not read");
            c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
            c.bw = (c.dw + 7) / 8; c.bh = (c.dh + 7) / 8;
            c.pw = mcux * c.h; c.ph = mcuy * c.v;
            c.coef.assign((size_t)c.pw * c.ph * 64, 0);
        }
        progressive = prog; have_frame = true;
        return true;
    }

    // ---- one scan ----
    struct Scan { int n = 0, ci[3] = {0, 0, 0}, Ss = 0, Se = 63, Ah = 0, Al = 0; };

    void decode_block_sequential(BitReader &br, Component &c, short *b) {
        int s = br.decode(dc[c.dc_tbl]);
        if (s > 16) s = 16;
        const int diff = s ? extend(br.bits(s), s) : 0;
        c.pred = (int)((unsigned)c.pred + (unsigned)diff);
        b[0] = (short)c.pred;
        const Huffman &h = ac[c.ac_tbl];
        for (int k = 1; k < 64; k++) {
            const int rs = br.decode(h), r = rs >> 4; s = rs & 15;
            if (s) { k += r; b[kNatural[k]] = (short)extend(br.bits(s), s); }
            else if (r == 15) k += 15;
            else break;
        }
    }
    void refine_nonzero(BitReader &br, short *coef, int p1, int m1) {
        if (br.bit() && (*coef & p1) == 0) *coef = (short)(*coef >= 0 ? *coef + p1 : *coef + m1);
    }
    void decode_block_progressive(BitReader &br, Component &c, short *b, const Scan &sc, unsigned &eobrun) {
        const int Al = sc.Al;
        if (sc.Ss == 0) {                                           // DC
            if (sc.Ah == 0) {
                int s = br.decode(dc[c.dc_tbl]);
                if (s > 16) s = 16;
                c.pred = (int)((unsigned)c.pred + (unsigned)(s ? extend(br.bits(s), s) : 0));
                b[0] = (short)((unsigned)c.pred << Al);
            } else if (br.bit()) b[0] = (short)(b[0] | (1 << Al));
            return;
        }
        const Huffman &h = ac[c.ac_tbl];
        if (sc.Ah == 0) {                                           // AC, first pass of the band
            if (eobrun) { eobrun--; return; }
            for (int k = sc.Ss; k <= sc.Se; k++) {
                const int rs = br.decode(h), r = rs >> 4, s = rs & 15;
                if (s) { k += r; b[kNatural[k]] = (short)((unsigned)extend(br.bits(s), s) << Al); }
                else if (r == 15) k += 15;
                else { eobrun = 1u << r; if (r) eobrun += (unsigned)br.bits(r); eobrun--; break; }
            }
            return;
        }
        const int p1 = 1 << Al, m1 = -(1 << Al);                    // AC, refinement
        int k = sc.Ss;
        if (!eobrun) {
            for (; k <= sc.Se; k++) {
                const int rs = br.decode(h);
                int r = rs >> 4, s = rs & 15;
                if (s) s = br.bit() ? p1 : m1;                      // (a size other than 1 is corrupt data: treated as 1, as libjpeg does)
                else if (r != 15) { eobrun = 1u << r; if (r) eobrun += (unsigned)br.bits(r); break; }
                do {
                    short *t = b + kNatural[k];
                    if (*t) refine_nonzero(br, t, p1, m1);
                    else if (--r < 0) break;
                    k++;
                } while (k <= sc.Se);
                if (s && k <= sc.Se) b[kNatural[k]] = (short)s;
            }
        }
        if (eobrun) {
            for (; k <= sc.Se; k++) { short *t = b + kNatural[k]; if (*t) refine_nonzero(br, t, p1, m1); }
            eobrun--;
        }
    }

    // entropy-coded data of one scan from `p`; returns the position of the marker that ends it
    const unsigned char *read_scan(const unsigned char *p, const unsigned char *end, const Scan &sc) {
        BitReader br; br.start(p, end);
        unsigned eobrun = 0;
        for (int i = 0; i < sc.n; i++) comp[sc.ci[i]].pred = 0;
        const bool single = sc.n == 1;
        const int mcux = single ? comp[sc.ci[0]].bw : (W + 8 * hmax - 1) / (8 * hmax), mcuy = single ? comp[sc.ci[0]].bh : (H + 8 * vmax - 1) / (8 * vmax);
        int until_restart = restart_interval, next_rst = 0;
        for (int my = 0; my < mcuy; my++) {
            for (int mx = 0; mx < mcux; mx++) {
                if (restart_interval && until_restart == 0) {
                    br.align();
                    const unsigned char *q = br.p;                  // the RSTn marker (fill bytes allowed in front)
                    while (q + 1 < end && q[0] == 0xFF && q[1] == 0xFF) q++;
                    if (q + 1 < end && q[0] == 0xFF && q[1] == (0xD0 | next_rst)) { q += 2; br.start(q, end); }
                    else if (!br.hit_marker) {                      // out of step: look for the next marker, as a resynchronisation would
                        while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF)) q++;
                        if (q + 1 < end && q[1] == (0xD0 | next_rst)) q += 2;
                        br.start(q, end);
                    } else br.start(q, end);
                    next_rst = (next_rst + 1) & 7;
                    until_restart = restart_interval; eobrun = 0;
                    for (int i = 0; i < sc.n; i++) comp[sc.ci[i]].pred = 0;
                }
                for (int i = 0; i < sc.n; i++) {
                    Component &c = comp[sc.ci[i]];
                    const int nh = single ? 1 : c.h, nv = single ? 1 : c.v;
                    for (int by = 0; by < nv; by++)
                        for (int bx = 0; bx < nh; bx++) {
                            const int X = mx * nh + bx, Y = my * nv + by;      // < pw, ph by construction
                            short *b = &c.coef[((size_t)Y * c.pw + X) * 64];
                            if (progressive) decode_block_progressive(br, c, b, sc, eobrun);
                            else decode_block_sequential(br, c, b);
                        }
                }
                if (restart_interval) until_restart--;
            }
        }
        const unsigned char *q = br.p;                              // skip to the marker that follows (bytes not consumed are padding)
        while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        return q;
    }

    bool read_sos(const unsigned char *d, int len, Scan &sc) {
        if (!have_frame) return fail("scan before the frame header");
        if (len < 1) return fail("bad SOS");
        sc.n = d[0];
        if (sc.n < 1 || sc.n > ncomp || len < 1 + 2 * sc.n + 3) return fail("bad SOS");
        for (int i = 0; i < sc.n; i++) {
            int ci = -1;
            for (int k = 0; k < ncomp; k++) if (comp[k].id == d[1 + 2 * i]) { ci = k; break; }
            if (ci < 0) return fail("scan names an unknown component");
            for (int k = 0; k < i; k++) if (sc.ci[k] == ci) return fail("scan names a component twice");
            sc.ci[i] = ci;
            comp[ci].dc_tbl = d[2 + 2 * i] >> 4; comp[ci].ac_tbl = d[2 + 2 * i] & 15;
            if (comp[ci].dc_tbl > 3 || comp[ci].ac_tbl > 3) return fail("bad table selector");
        }
        const unsigned char *t = d + 1 + 2 * sc.n;
        sc.Ss = t[0]; sc.Se = t[1]; sc.Ah = t[2] >> 4; sc.Al = t[2] & 15;
        if (progressive) {
            if (sc.Ss > sc.Se || sc.Se > 63 || sc.Al > 13 || sc.Ah > 13) return fail("bad progression parameters");
            if (sc.Ss == 0 ? sc.Se != 0 : sc.n != 1) return fail("bad progression parameters");
        } else { sc.Ss = 0; sc.Se = 63; sc.Ah = sc.Al = 0; }
        if (sc.n > 1) {                                             // blocks per MCU: at most 10 (T.81 B.2.3)
            int blocks = 0;
            for (int i = 0; i < sc.n; i++) blocks += comp[sc.ci[i]].h * comp[sc.ci[i]].v;
            if (blocks > 10) return fail("too many blocks per MCU");
        }
        for (int i = 0; i < sc.n; i++) {
            Component &c = comp[sc.ci[i]];
            const bool need_dc = sc.Ss == 0 && sc.Ah == 0, need_ac = progressive ? sc.Ss > 0 : true;
            if (need_dc && !dc[c.dc_tbl].defined) return fail("scan uses an undefined DC table");
            if (need_ac && !ac[c.ac_tbl].defined) return fail("scan uses an undefined AC table");
            if (!c.latched) {
                if (!qt_defined[c.tq]) return fail("component uses an undefined quantisation table");
                std::memcpy(c.quant, qt[c.tq], sizeof c.quant); c.latched = true;
            }
        }
        return true;
    }

    bool parse(const std::vector<unsigned char> &file) {
        const unsigned char *p = file.data(), *end = p + file.size();
        file_size = file.size();
        if (file.size() < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG");
        p += 2;
        bool seen_scan = false;
        while (true) {
            while (p < end && *p != 0xFF) p++;                      // (garbage between segments is skipped, as libjpeg does with a warning)
            while (p < end && *p == 0xFF) p++;
            if (p >= end) break;
            const int m = *p++;
            if (m == 0xD9) break;                                   // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;    // TEM, stray RSTn: no payload
            if (p + 2 > end) return fail("truncated segment");
            const int len = ((p[0] << 8) | p[1]) - 2;
            if (len < 0 || p + 2 + len > end) return fail("truncated segment");
            const unsigned char *d = p + 2;
            p += 2 + len;
            if (m == 0xDB) { if (!read_dqt(d, len)) return false; }
            else if (m == 0xC4) { if (!read_dht(d, len)) return false; }
            else if (m == 0xC0 || m == 0xC1) { if (!read_sof(d, len, false)) return false; }
            else if (m == 0xC2) { if (!read_sof(d, len, true)) return false; }
            else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) return fail("lossless, hierarchical and arithmetic-coded JPEGs are not read");
            else if (m == 0xDD) { if (len < 2) return fail("bad DRI"); restart_interval = (d[0] << 8) | d[1]; }
            else if (m == 0xE0) { if (len >= 5 && !std::memcmp(d, "JFIF", 5)) jfif = true; }
            else if (m == 0xEE) { if (len >= 12 && !std::memcmp(d, "Adobe", 5)) { adobe = true; adobe_transform = d[11]; } }
            else if (m == 0xDA) {
                Scan sc;
                if (!read_sos(d, len, sc)) return false;
                p = read_scan(p, end, sc);
                seen_scan = true;
            }
        }
        if (!have_frame || !seen_scan) return fail("no image data");
        return true;
    }

    // ---- samples: inverse DCT of every stored block, upsampling, colour ----
    void reconstruct() {
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            const int stride = 8 * c.pw;
            c.plane.assign((size_t)stride * 8 * c.ph, 0);
            for (int by = 0; by < c.ph; by++)
                for (int bx = 0; bx < c.pw; bx++) idct(&c.coef[((size_t)by * c.pw + bx) * 64], c.quant, &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
        }
    }
    // row y of the full-size image of component c into out[0 .. >= W)
    void upsampled_row(const Component &c, int y, unsigned char *out, std::vector<int> &tmp) const {
        const int stride = 8 * c.pw, fx = hmax / c.h, fy = vmax / c.v;
        const bool fancy = c.dw > 2;                                // (libjpeg filters only components more than two samples wide)
        auto row = [&](int r) { return &c.plane[(size_t)(r < 0 ? 0 : (r >= c.dh ? c.dh - 1 : r)) * stride]; };
        if (fx == 1 && fy == 1) { std::memcpy(out, row(y), (size_t)W); return; }
        if (fancy && fx == 2 && fy == 1) {                          // h2v1: 3/4 of the nearer, 1/4 of the further sample
            const unsigned char *in = row(y);
            const int n = c.dw;
            for (int x = 0; x < n; x++) {
                const int v = in[x] * 3;
                out[2 * x] = x == 0 ? in[0] : (unsigned char)((v + in[x - 1] + 1) >> 2);
                out[2 * x + 1] = x == n - 1 ? in[x] : (unsigned char)((v + in[x + 1] + 2) >> 2);
            }
            return;
        }
        if (fx == 1 && fy == 2) {                          // h1v2
            const int r = y >> 1, up = !(y & 1);
            const unsigned char *a = row(r), *b = row(up ? r - 1 : r + 1);
            for (int x = 0; x < c.dw; x++) out[x] = (unsigned char)((a[x] * 3 + b[x] + (up ? 1 : 2)) >> 2);
            return;
        }
        if (fancy && fx == 2 && fy == 2) {                          // h2v2: the same filter in both directions, 9 3 3 1 / 16
            const int r = y >> 1, up = !(y & 1);
            const unsigned char *a = row(r), *b = row(up ? r - 1 : r + 1);
            const int n = c.dw;
            tmp.resize((size_t)n);
            for (int x = 0; x < n; x++) tmp[x] = a[x] * 3 + b[x];
            for (int x = 0; x < n; x++) {
                out[2 * x] = (unsigned char)(x == 0 ? (tmp[0] * 4 + 8) >> 4 : (tmp[x] * 3 + tmp[x - 1] + 8) >> 4);
                out[2 * x + 1] = (unsigned char)(x == n - 1 ? (tmp[x] * 4 + 7) >> 4 : (tmp[x] * 3 + tmp[x + 1] + 7) >> 4);
            }
            return;
        }
        const unsigned char *in = row(y / fy);                      // any other (integral) ratio: replication
        for (int x = 0; x < W; x++) out[x] = in[x / fx];
    }
};

// RGB (three channels) or gray (one) into `px`, row-major, top to bottom
static inline bool decode(const std::vector<unsigned char> &file, int &w, int &h, int &ch, std::vector<unsigned char> &px, std::string *why = nullptr) {
    Decoder d;
    auto refuse = [&](const char *msg) { if (why) *why = msg; return false; };
    if (!d.parse(file)) return refuse(d.error.c_str());
    for (int i = 0; i < d.ncomp; i++) if (!d.comp[i].latched) return refuse("a component has no scan");
    d.reconstruct();
    w = d.W; h = d.H; ch = d.ncomp == 1 ? 1 : 3;
    px.resize((size_t)w * h * ch);
    // three components: YCbCr unless the file says otherwise (JFIF: always; Adobe: its transform flag; neither: component ids 'R' 'G' 'B' mean RGB)
    bool ycc = true;
    if (d.ncomp == 3 && !d.jfif) {
        if (d.adobe) ycc = d.adobe_transform != 0;
        else if (d.comp[0].id == 'R' && d.comp[1].id == 'G' && d.comp[2].id == 'B') ycc = false;
    }
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; i++) {
        const int x = i - 128;
        cr_r[i] = (91881 * x + 32768) >> 16;                        // 1.40200
        cb_b[i] = (116130 * x + 32768) >> 16;                       // 1.77200
        cr_g[i] = -46802 * x;                                       // 0.71414
        cb_g[i] = -22554 * x + 32768;                               // 0.34414, with the rounding of the sum
    }
    auto clamp = [](int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    size_t maxw = (size_t)w + 16;
    for (int i = 0; i < d.ncomp; i++) { const size_t n = (size_t)16 * d.comp[i].pw + 16; if (n > maxw) maxw = n; }
    std::vector<unsigned char> r0(maxw), r1(maxw), r2(maxw);
    std::vector<int> tmp;
    for (int y = 0; y < h; y++) {
        unsigned char *o = &px[(size_t)y * w * ch];
        d.upsampled_row(d.comp[0], y, r0.data(), tmp);
        if (ch == 1) { std::memcpy(o, r0.data(), (size_t)w); continue; }
        d.upsampled_row(d.comp[1], y, r1.data(), tmp);
        d.upsampled_row(d.comp[2], y, r2.data(), tmp);
        for (int x = 0; x < w; x++) {
            if (ycc) {
                const int Y = r0[x], cb = r1[x], cr = r2[x];
                o[3 * x] = clamp(Y + cr_r[cr]);
                o[3 * x + 1] = clamp(Y + ((cb_g[cb] + cr_g[cr]) >> 16));
                o[3 * x + 2] = clamp(Y + cb_b[cb]);
            } else { o[3 * x] = r0[x]; o[3 * x + 1] = r1[x]; o[3 * x + 2] = r2[x]; }
        }
    }
    return true;
}

static inline bool read_file(const std::string &path, int &w, int &h, int &ch, std::vector<unsigned char> &px, std::string *why = nullptr) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { if (why) *why = "cannot open"; return false; }
    std::vector<unsigned char> file;
    unsigned char buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    return decode(file, w, h, ch, px, why);
}

}  // namespace rtdd_jpeg
