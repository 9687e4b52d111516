// y2_codec.cpp -- the host's own JPEG and PNG decoders.
//
// The reference host opens its input through load_image_stb (src/core/yolo_image.cpp:167-189): stbi_load(file, 3 channels)
// from its vendored stb_image v2.19 (src/stb_image_implementation.cpp), then bytes / 255.  A detector fed by another decoder sees
// other pixels (PIL's JPEG output differs from stb's in the last bit of many samples), so this file decodes to the SAME bytes:
// the lossless parts (Huffman / inflate, PNG filters, Adam7) are the standards', the lossy or lossy-looking parts follow the
// arithmetic stb fixes -
//   * dequantised coefficients are kept in 16 bits (value * q truncated to int16);
//   * inverse DCT: the IJG "islow" factorisation in 12-bit fixed point, columns first with 2 extra bits kept (+512 >> 10; a column
//     whose AC terms are all zero is its DC term * 4), then rows with +65536 + (128 << 17) >> 17, clamped to 0..255;
//   * chroma upsampling: centred triangle filters - (3 near + far + 2) >> 2 in one direction, (3 a + b + 8) >> 4 of the vertical
//     sums in both, edge samples replicated; any other ratio by sample replication;
//   * YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits (the Cb term of green masked to its upper 16 bits);
//   * PNG: 16-bit samples keep their high byte, 1/2/4-bit grey samples are multiplied by 255 / 85 / 17, alpha is dropped.
// Baseline / extended-sequential and progressive Huffman JPEG (8-bit; 1, 3 or 4 components; any sampling factors 1..4; restart
// intervals; Adobe transform flag), every PNG colour type and bit depth, interlaced or not.  Checked byte for byte against what the
// compiled reference's stb produces for its nine example images and for a set of synthetic encodings of every supported variant
// (tests/golden/images.npz, tests/test_host_codec.py).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "y2_host.hpp"

namespace y2h {
namespace {

[[noreturn]] void bad(const std::string &what) { throw std::runtime_error(what); }

// ============================================================================================== JPEG

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Canonical Huffman table (ITU T.81 Annex C / F.2.2.3): codes of length L are consecutive, starting at mincode[L].
struct Huff {
    int maxcode[18];      // largest code of each length, -1 if none; [17] = sentinel
    int valptr[17], mincode[17];
    uint8_t vals[256];
    uint8_t look_len[512], look_val[512];   // 9-bit prefix -> (length, symbol) for codes of <= 9 bits
    bool defined = false;
    void build(const int counts[16], const uint8_t *symbols, int n)
    {
        memcpy(vals, symbols, (size_t)n);
        memset(look_len, 0, sizeof(look_len));
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code)
                if (len <= 9) {
                    const int first = code << (9 - len), span = 1 << (9 - len);
                    if (first + span > 512) bad("Corrupt JPEG: bad code lengths");
                    for (int j = 0; j < span; ++j) { look_len[first + j] = (uint8_t)len; look_val[first + j] = symbols[k]; }
                }
            if (code > (1 << len)) bad("Corrupt JPEG: bad code lengths");
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
    }
};

// MSB-first bit reader over an entropy-coded segment.  0xFF00 is a data byte 0xFF; 0xFF followed by anything else is a marker:
// reading stops there and the decoder is fed zero bits from then on (what a truncated scan decodes to is then defined).
struct Bits {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    int marker = -1;      // marker met inside / after the segment
    bool dry = false;
    void reset() { acc = 0; n = 0; marker = -1; dry = false; }
    void fill()
    {
        while (n <= 24) {
            uint32_t b = 0;
            if (!dry) {
                b = p < end ? *p++ : 0;
                if (b == 0xff) {
                    int c = p < end ? *p++ : 0;
                    while (c == 0xff) c = p < end ? *p++ : 0;     // fill bytes
                    if (c != 0) { marker = c; dry = true; b = 0; }
                }
            }
            acc |= b << (24 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)(acc >> (32 - k)); }
    void drop(int k) { acc <<= k; n -= k; }
    int get(int k) { if (k == 0) return 0; if (n < k) fill(); const int v = (int)(acc >> (32 - k)); drop(k); return v; }
    int bit() { return get(1); }
    int symbol(const Huff &h)
    {
        if (n < 16) fill();
        const int pre = (int)(acc >> 23);
        if (h.look_len[pre]) { drop(h.look_len[pre]); return h.look_val[pre]; }
        int code = (int)(acc >> 22), len = 10;                       // no code of <= 9 bits matched
        for (; len <= 16; ++len, code = (int)(acc >> (32 - len)))
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) break;
        if (len > 16) bad("Corrupt JPEG: bad huffman code");
        drop(len);
        return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    // RECEIVE + EXTEND (T.81 F.2.2.1): k magnitude bits, values with a leading 0 bit are negative
    int extend(int k)
    {
        if (k == 0) return 0;
        const int v = get(k);
        return v < (1 << (k - 1)) ? v - (1 << k) + 1 : v;
    }
};

constexpr int fx12(float x) { return (int)(x * 4096 + 0.5); }

// One 1-D pass of the inverse DCT on s[0..7]; returns the even part in x[0..3] and the odd part in t[0..3] such that the outputs
// are x0 +- t3, x1 +- t2, x2 +- t1, x3 +- t0 (scaled by 4096).
inline void idct_1d(const int s[8], int x[4], int t[4])
{
    int p2 = s[2], p3 = s[6];
    int p1 = (p2 + p3) * fx12(0.5411961f);
    const int e2 = p1 + p3 * fx12(-1.847759065f), e3 = p1 + p2 * fx12(0.765366865f);
    p2 = s[0]; p3 = s[4];
    const int e0 = (p2 + p3) * 4096, e1 = (p2 - p3) * 4096;
    x[0] = e0 + e3; x[3] = e0 - e3; x[1] = e1 + e2; x[2] = e1 - e2;
    int t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
    p3 = t0 + t2;
    int p4 = t1 + t3;
    p1 = t0 + t3;
    p2 = t1 + t2;
    const int p5 = (p3 + p4) * fx12(1.175875602f);
    t0 *= fx12(0.298631336f); t1 *= fx12(2.053119869f); t2 *= fx12(3.072711026f); t3 *= fx12(1.501321110f);
    p1 = p5 + p1 * fx12(-0.899976223f);
    p2 = p5 + p2 * fx12(-2.562915447f);
    p3 *= fx12(-1.961570560f);
    p4 *= fx12(-0.390180644f);
    t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
}

inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void idct_block(uint8_t *out, int stride, const int16_t d[64])
{
    int v[64];
    for (int c = 0; c < 8; ++c) {
        if (!d[c + 8] && !d[c + 16] && !d[c + 24] && !d[c + 32] && !d[c + 40] && !d[c + 48] && !d[c + 56]) {
            const int dc = d[c] * 4;
            for (int r = 0; r < 8; ++r) v[r * 8 + c] = dc;
            continue;
        }
        const int s[8] = {d[c], d[c + 8], d[c + 16], d[c + 24], d[c + 32], d[c + 40], d[c + 48], d[c + 56]};
        int x[4], t[4];
        idct_1d(s, x, t);
        for (int k = 0; k < 4; ++k) {
            const int e = x[k] + 512;
            v[k * 8 + c] = (e + t[3 - k]) >> 10;
            v[(7 - k) * 8 + c] = (e - t[3 - k]) >> 10;
        }
    }
    for (int r = 0; r < 8; ++r, out += stride) {
        int x[4], t[4];
        idct_1d(v + r * 8, x, t);
        for (int k = 0; k < 4; ++k) {
            const int e = x[k] + 65536 + (128 << 17);
            out[k] = clamp255((e + t[3 - k]) >> 17);
            out[7 - k] = clamp255((e - t[3 - k]) >> 17);
        }
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int x = 0, y = 0;          // effective pixels
    int w2 = 0, h2 = 0;        // allocated plane (whole MCUs)
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;   // progressive: [(h2/8) * (w2/8)][64], natural order
};

struct Jpeg {
    const uint8_t *p, *end;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int restart_interval = 0;
    uint16_t dq[4][64] = {};
    Huff hdc[4], hac[4];
    Component comp[4];
    int hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0;
    // current scan
    int scan_n = 0, order[4] = {0, 0, 0, 0}, ss = 0, se = 63, ah = 0, al = 0, eob_run = 0, todo = 0;
    Bits br;

    int u8() { return p < end ? *p++ : 0; }
    int u16() { const int a = u8(); return (a << 8) | u8(); }
    void skip(int n) { p = (n > end - p) ? end : p + n; }
    bool eof() const { return p >= end; }

    int next_marker()
    {
        if (br.marker >= 0) { const int m = br.marker; br.marker = -1; return m; }
        int x = u8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = u8();
        return x;
    }

    void segment(int m)
    {
        switch (m) {
        case -1: bad("Corrupt JPEG: expected marker");
        case 0xDD:
            if (u16() != 4) bad("Corrupt JPEG: bad DRI len");
            restart_interval = u16();
            return;
        case 0xDB: {
            int L = u16() - 2;
            while (L > 0) {
                const int q = u8(), wide = q >> 4, t = q & 15;
                if (wide > 1) bad("Corrupt JPEG: bad DQT type");
                if (t > 3) bad("Corrupt JPEG: bad DQT table");
                for (int i = 0; i < 64; ++i) dq[t][kZigzag[i]] = (uint16_t)(wide ? u16() : u8());
                L -= wide ? 129 : 65;
            }
            if (L != 0) bad("Corrupt JPEG: bad DQT len");
            return;
        }
        case 0xC4: {
            int L = u16() - 2;
            while (L > 0) {
                const int q = u8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) bad("Corrupt JPEG: bad DHT header");
                int counts[16], n = 0;
                for (int &c : counts) { c = u8(); n += c; }
                if (n > 256) bad("Corrupt JPEG: bad DHT header");
                uint8_t syms[256];
                for (int i = 0; i < n; ++i) syms[i] = (uint8_t)u8();
                (tc ? hac : hdc)[th].build(counts, syms, n);
                L -= 17 + n;
            }
            if (L != 0) bad("Corrupt JPEG: bad DHT len");
            return;
        }
        default: break;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            int L = u16();
            if (L < 2) bad(m == 0xFE ? "Corrupt JPEG: bad COM len" : "Corrupt JPEG: bad APP len");
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) ok &= u8() == tag[i];
                L -= 5;
                if (ok) jfif = true;
            } else if (m == 0xEE && L >= 12) {
                static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) ok &= u8() == tag[i];
                L -= 6;
                if (ok) { u8(); u16(); u16(); adobe_transform = u8(); L -= 6; }
            }
            skip(L);
            return;
        }
        bad("Corrupt JPEG: unknown marker");
    }

    void frame_header()
    {
        const int Lf = u16();
        if (Lf < 11) bad("Corrupt JPEG: bad SOF len");
        if (u8() != 8) bad("JPEG format not supported: 8-bit only");
        height = u16();
        if (!height) bad("JPEG format not supported: delayed height");
        width = u16();
        if (!width) bad("Corrupt JPEG: 0 width");
        ncomp = u8();
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) bad("Corrupt JPEG: bad component count");
        if (Lf != 8 + 3 * ncomp) bad("Corrupt JPEG: bad SOF len");
        if ((long long)width * height * ncomp > (1LL << 30)) bad("Image too large to decode");
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.id = u8();
            if (ncomp == 3 && c.id == "RGB"[i]) ++rgb_ids;
            const int q = u8();
            c.h = q >> 4; c.v = q & 15;
            if (!c.h || c.h > 4) bad("Corrupt JPEG: bad H");
            if (!c.v || c.v > 4) bad("Corrupt JPEG: bad V");
            c.tq = u8();
            if (c.tq > 3) bad("Corrupt JPEG: bad TQ");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        // sampling factors that do not divide the largest one (H = 4,3,1; 3,2,1): the up-sampler's integer ratio hmax / h would be
        // too small and a row of `width` samples would be read from a plane row of only mcu_x * h * 8 (heap read past
        // the plane under ASan).  stb 2.19 survives such files on 15 bytes of over-allocation, later stb versions reject them; so do we.
        for (int i = 0; i < ncomp; ++i)
            if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0) bad("Corrupt JPEG: bad H/V");
        mcu_x = (width + hmax * 8 - 1) / (hmax * 8);
        mcu_y = (height + vmax * 8 - 1) / (vmax * 8);
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax;
            c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
        }
    }

    void scan_header()
    {
        const int Ls = u16();
        scan_n = u8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp) bad("Corrupt JPEG: bad SOS component count");
        if (Ls != 6 + 2 * scan_n) bad("Corrupt JPEG: bad SOS len");
        for (int i = 0; i < scan_n; ++i) {
            const int id = u8(), q = u8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) ++which;
            if (which == ncomp) bad("Corrupt JPEG: bad SOS component");
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3) bad("Corrupt JPEG: bad DC huff");
            if (comp[which].ha > 3) bad("Corrupt JPEG: bad AC huff");
            order[i] = which;
        }
        ss = u8(); se = u8();
        const int a = u8();
        ah = a >> 4; al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) bad("Corrupt JPEG: bad SOS");
        } else {
            if (ss != 0 || ah != 0 || al != 0) bad("Corrupt JPEG: bad SOS");
            se = 63;
        }
    }

    void restart()
    {
        br.reset();
        for (Component &c : comp) c.dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    // ---- sequential: one block = DC difference + AC run/size pairs, dequantised into 16 bits as they are decoded
    void block_sequential(Component &c, int16_t d[64])
    {
        const Huff &hd = hdc[c.hd], &ha = hac[c.ha];
        if (!hd.defined || !ha.defined) bad("Corrupt JPEG: missing huffman table");
        const uint16_t *q = dq[c.tq];
        memset(d, 0, 64 * sizeof(int16_t));
        const int t = br.symbol(hd);
        if (t > 16) bad("Corrupt JPEG: bad huffman code");
        c.dc_pred += br.extend(t);
        d[0] = (int16_t)(c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = br.symbol(ha), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) bad("Corrupt JPEG: bad huffman code");
                const int z = kZigzag[k++];
                d[z] = (int16_t)(br.extend(s) * q[z]);
            }
        }
    }

    // ---- progressive (T.81 Annex G): DC first / refinement
    void block_prog_dc(Component &c, int16_t d[64])
    {
        if (se != 0) bad("Corrupt JPEG: can't merge dc and ac");
        if (ah == 0) {
            const Huff &hd = hdc[c.hd];
            if (!hd.defined) bad("Corrupt JPEG: missing huffman table");
            memset(d, 0, 64 * sizeof(int16_t));
            const int t = br.symbol(hd);
            if (t > 16) bad("Corrupt JPEG: bad huffman code");
            c.dc_pred += br.extend(t);
            d[0] = (int16_t)((unsigned)c.dc_pred << al);   // (shifted as unsigned: the value may be negative; same low 16 bits)
        } else if (br.bit()) {
            d[0] = (int16_t)(d[0] + (1 << al));
        }
    }

    static void refine(int16_t &v, int bit)
    {
        if ((v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    }

    // AC first / refinement of coefficients ss..se of one block
    void block_prog_ac(Component &c, int16_t d[64])
    {
        if (ss == 0) bad("Corrupt JPEG: can't merge dc and ac");
        const Huff &ha = hac[c.ha];
        if (!ha.defined) bad("Corrupt JPEG: missing huffman table");
        if (ah == 0) {
            if (eob_run) { --eob_run; return; }
            for (int k = ss; k <= se;) {
                const int rs = br.symbol(ha), r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r) - 1;
                        if (r) eob_run += br.get(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) bad("Corrupt JPEG: bad huffman code");
                    d[kZigzag[k++]] = (int16_t)((unsigned)br.extend(s) << al);
                }
            }
            return;
        }
        const int bit = 1 << al;
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; ++k) {
                int16_t &v = d[kZigzag[k]];
                if (v != 0 && br.bit()) refine(v, bit);
            }
            return;
        }
        for (int k = ss; k <= se;) {
            const int rs = br.symbol(ha);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += br.get(r);
                    r = 64;      // the rest of the band: refinements only
                }
                // r == 15: sixteen zero-history coefficients are skipped (the 16th "receives" s = 0)
            } else {
                if (s != 1) bad("Corrupt JPEG: bad huffman code");
                s = br.bit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t &v = d[kZigzag[k++]];
                if (v != 0) {
                    if (br.bit()) refine(v, bit);
                } else {
                    if (r == 0) { v = (int16_t)s; break; }
                    --r;
                }
            }
        }
    }

    // returns false when a restart interval ended without a restart marker (the rest of the scan is then left as it is)
    bool interval_done()
    {
        if (--todo > 0) return true;
        if (br.n < 24) br.fill();
        if (br.marker < 0xd0 || br.marker > 0xd7) return false;
        restart();
        return true;
    }

    void scan()
    {
        restart();
        br.p = p; br.end = end;
        int16_t blk[64];
        auto one = [&](Component &c, int bx, int by) {
            if (!progressive) {
                block_sequential(c, blk);
                idct_block(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, blk);
            } else {
                int16_t *d = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
                if (ss == 0) block_prog_dc(c, d);
                else block_prog_ac(c, d);
            }
        };
        bool go = true;
        if (scan_n == 1) {     // non-interleaved: the component's own blocks in raster order
            Component &c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh && go; ++j)
                for (int i = 0; i < bw && go; ++i) {
                    one(c, i, j);
                    go = interval_done();
                }
        } else {
            if (progressive && ss != 0) bad("Corrupt JPEG: interleaved AC scan");
            for (int j = 0; j < mcu_y && go; ++j)
                for (int i = 0; i < mcu_x && go; ++i) {
                    for (int k = 0; k < scan_n; ++k) {
                        Component &c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) one(c, i * c.h + x, j * c.v + y);
                    }
                    go = interval_done();
                }
        }
        p = br.p;
    }

    void finish_progressive()
    {
        for (int n = 0; n < ncomp; ++n) {
            Component &c = comp[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            const uint16_t *q = dq[c.tq];
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    int16_t *d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (c.w2 / 8));
                    for (int k = 0; k < 64; ++k) d[k] = (int16_t)(d[k] * q[k]);
                    idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, d);
                }
        }
    }

    void decode()
    {
        if (next_marker() != 0xD8) bad("Corrupt JPEG: no SOI");
        int m = next_marker();
        while (m != 0xC0 && m != 0xC1 && m != 0xC2) {
            segment(m);
            m = next_marker();
            while (m == -1) {
                if (eof()) bad("Corrupt JPEG: no SOF");
                m = next_marker();
            }
        }
        progressive = m == 0xC2;
        frame_header();
        m = next_marker();
        while (m != 0xD9) {
            if (m == 0xDA) {
                scan_header();
                scan();
                if (br.marker < 0) {      // junk between the scan and the next marker
                    while (!eof()) {
                        if (u8() == 0xff) { br.marker = u8(); break; }
                    }
                }
            } else if (m == 0xDC) {
                const int Ld = u16(), NL = u16();
                if (Ld != 4) bad("Corrupt JPEG: bad DNL len");
                if (NL != height) bad("Corrupt JPEG: bad DNL height");
            } else {
                segment(m);
            }
            m = next_marker();
        }
        if (progressive) finish_progressive();
    }
};

// ---- chroma upsampling: one output row of `w` low-resolution samples expanded by hs horizontally
typedef const uint8_t *(*RowFn)(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int hs);

const uint8_t *row_1(uint8_t *, const uint8_t *near, const uint8_t *, int, int) { return near; }
const uint8_t *row_v2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int)
{
    for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
    return out;
}
const uint8_t *row_h2(uint8_t *out, const uint8_t *in, const uint8_t *, int w, int)
{
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < w - 1; ++i) {
        const int n = 3 * in[i] + 2;
        out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[(w - 1) * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[(w - 1) * 2 + 1] = in[w - 1];
    return out;
}
const uint8_t *row_hv2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int)
{
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return out; }
    int t1 = 3 * near[0] + far[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
        const int t0 = t1;
        t1 = 3 * near[i] + far[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}
const uint8_t *row_rep(uint8_t *out, const uint8_t *near, const uint8_t *, int w, int hs)
{
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
    return out;
}

constexpr int fx20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

inline void ycc_to_rgb(uint8_t *out, int y, int cb, int cr)
{
    const int yf = (y << 20) + (1 << 19);
    cr -= 128; cb -= 128;
    const int r = (yf + cr * fx20(1.40200f)) >> 20;
    const int g = (int)(yf + cr * -fx20(0.71414f) + (int)((unsigned)(cb * -fx20(0.34414f)) & 0xffff0000u)) >> 20;
    const int b = (yf + cb * fx20(1.77200f)) >> 20;
    out[0] = clamp255(r); out[1] = clamp255(g); out[2] = clamp255(b);
}

inline uint8_t mul255(int x, int y)   // x * y / 255, rounded
{
    const unsigned t = (unsigned)(x * y + 128);
    return (uint8_t)((t + (t >> 8)) >> 8);
}

ImageU8 jpeg_to_rgb(Jpeg &j)
{
    ImageU8 im;
    im.w = j.width; im.h = j.height;
    im.rgb.resize((size_t)j.width * j.height * 3);
    struct Up {
        RowFn fn;
        const uint8_t *line0, *line1;
        int hs, vs, w_lores, ystep, ypos;
        std::vector<uint8_t> buf;
    } up[4];
    for (int k = 0; k < j.ncomp; ++k) {
        Component &c = j.comp[k];
        Up &u = up[k];
        u.hs = j.hmax / c.h; u.vs = j.vmax / c.v;
        u.ystep = u.vs >> 1;
        u.w_lores = (j.width + u.hs - 1) / u.hs;
        u.ypos = 0;
        u.line0 = u.line1 = c.plane.data();
        u.buf.assign((size_t)j.width + 3 + 8, 0);
        u.fn = (u.hs == 1 && u.vs == 1) ? row_1 : (u.hs == 1 && u.vs == 2) ? row_v2 : (u.hs == 2 && u.vs == 1) ? row_h2 : (u.hs == 2 && u.vs == 2) ? row_hv2 : row_rep;
    }
    const bool is_rgb = j.ncomp == 3 && (j.rgb_ids == 3 || (j.adobe_transform == 0 && !j.jfif));
    for (int y = 0; y < j.height; ++y) {
        const uint8_t *row[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < j.ncomp; ++k) {
            Up &u = up[k];
            const bool bot = u.ystep >= (u.vs >> 1);
            row[k] = u.fn(u.buf.data(), bot ? u.line1 : u.line0, bot ? u.line0 : u.line1, u.w_lores, u.hs);
            if (++u.ystep >= u.vs) {
                u.ystep = 0;
                u.line0 = u.line1;
                if (++u.ypos < j.comp[k].y) u.line1 += j.comp[k].w2;
            }
        }
        uint8_t *out = im.rgb.data() + (size_t)y * j.width * 3;
        for (int x = 0; x < j.width; ++x, out += 3) {
            if (j.ncomp == 1) {
                out[0] = out[1] = out[2] = row[0][x];
            } else if (j.ncomp == 3) {
                if (is_rgb) { out[0] = row[0][x]; out[1] = row[1][x]; out[2] = row[2][x]; }
                else ycc_to_rgb(out, row[0][x], row[1][x], row[2][x]);
            } else if (j.adobe_transform == 0) {          // CMYK
                const int m = row[3][x];
                out[0] = mul255(row[0][x], m); out[1] = mul255(row[1][x], m); out[2] = mul255(row[2][x], m);
            } else if (j.adobe_transform == 2) {          // YCCK
                ycc_to_rgb(out, row[0][x], row[1][x], row[2][x]);
                const int m = row[3][x];
                out[0] = mul255(255 - out[0], m); out[1] = mul255(255 - out[1], m); out[2] = mul255(255 - out[2], m);
            } else {
                ycc_to_rgb(out, row[0][x], row[1][x], row[2][x]);
            }
        }
    }
    return im;
}

// ============================================================================================== zlib inflate (RFC 1950 / 1951)

struct ZHuff {
    uint16_t count[16], first_code[17], first_sym[16];
    uint16_t sym[288];
    void build(const uint8_t *len, int n)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) ++count[len[i]];
        count[0] = 0;
        int code = 0, k = 0;
        uint16_t next[16];
        for (int l = 1; l < 16; ++l) {
            first_code[l] = (uint16_t)code;
            first_sym[l] = next[l] = (uint16_t)k;
            code += count[l];
            if (count[l] && code - 1 >= (1 << l)) bad("Corrupt PNG: bad code lengths");
            k += count[l];
            code <<= 1;
        }
        for (int i = 0; i < n; ++i)
            if (len[i]) sym[next[len[i]]++] = (uint16_t)i;
    }
};

struct Inflate {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    std::vector<uint8_t> out;
    size_t limit = (size_t)-1;   // bytes the caller can use: decoding stops there (a damaged stream cannot inflate without bound)
    int past_end = 0;            // zero bytes supplied after the end of the input (the reference's loader feeds zeros too)
    int bits(int k)
    {
        while (n < k) {
            if (p >= end && ++past_end > 8) bad("Corrupt PNG: zlib stream ends early");
            acc |= (uint32_t)(p < end ? *p++ : 0) << n; n += 8;
        }
        const int v = (int)(acc & ((1u << k) - 1));
        acc >>= k; n -= k;
        return v;
    }
    int decode(const ZHuff &h)
    {
        int code = 0;
        for (int l = 1; l < 16; ++l) {
            code = (code << 1) | bits(1);       // Huffman codes are packed MSB first
            const int off = code - h.first_code[l];
            if (off >= 0 && off < h.count[l]) return h.sym[h.first_sym[l] + off];
        }
        bad("Corrupt PNG: bad huffman code");
    }
    void run(bool header)
    {
        static const int len_base[31] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0};
        static const int len_extra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
        static const int dist_base[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0};
        static const int dist_extra[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
        if (header) {
            const int cmf = bits(8), flg = bits(8);
            if ((cmf * 256 + flg) % 31 != 0) bad("Corrupt PNG: bad zlib header");
            if (flg & 32) bad("Corrupt PNG: no preset dict");
            if ((cmf & 15) != 8) bad("Corrupt PNG: bad compression");
        }
        for (bool last = false; !last;) {
            last = bits(1) != 0;
            const int type = bits(2);
            if (type == 0) {
                bits(n & 7);     // to the byte boundary
                uint8_t hdr[4];
                for (uint8_t &b : hdr) b = (uint8_t)bits(8);
                const int len = hdr[0] | (hdr[1] << 8), nlen = hdr[2] | (hdr[3] << 8);
                if (nlen != (len ^ 0xffff)) bad("Corrupt PNG: zlib corrupt");
                for (int i = 0; i < len; ++i) out.push_back((uint8_t)bits(8));
                if (out.size() >= limit) return;
                continue;
            }
            if (type == 3) bad("Corrupt PNG: bad block type");
            ZHuff lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 288; ++i) lens[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
                lit.build(lens, 288);
                for (int i = 0; i < 32; ++i) lens[i] = 5;
                dist.build(lens, 32);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int hlit = bits(5) + 257, hdist = bits(5) + 1, hclen = bits(4) + 4;
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)bits(3);
                ZHuff clh;
                clh.build(cl, 19);
                int k = 0;
                while (k < hlit + hdist) {
                    const int c = decode(clh);
                    if (c < 16) { lens[k++] = (uint8_t)c; continue; }
                    int rep, fill = 0;
                    if (c == 16) {
                        if (k == 0) bad("Corrupt PNG: bad codelengths");
                        rep = 3 + bits(2); fill = lens[k - 1];
                    } else if (c == 17) rep = 3 + bits(3);
                    else rep = 11 + bits(7);
                    if (k + rep > hlit + hdist) bad("Corrupt PNG: bad codelengths");
                    while (rep--) lens[k++] = (uint8_t)fill;
                }
                lit.build(lens, hlit);
                dist.build(lens + hlit, hdist);
            }
            for (;;) {
                const int s = decode(lit);
                if (s < 256) {
                    out.push_back((uint8_t)s);
                    if (out.size() >= limit) return;
                    continue;
                }
                if (s == 256) break;
                if (s > 285) bad("Corrupt PNG: bad huffman code");
                const int len = len_base[s - 257] + bits(len_extra[s - 257]);
                const int d = decode(dist);
                if (d > 29) bad("Corrupt PNG: bad huffman code");
                const size_t back = (size_t)dist_base[d] + (size_t)bits(dist_extra[d]);
                if (back > out.size()) bad("Corrupt PNG: bad dist");
                const size_t from = out.size() - back;
                for (int i = 0; i < len; ++i) out.push_back(out[from + (size_t)i]);
                if (out.size() >= limit) return;
            }
        }
    }
};

// ============================================================================================== PNG

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// reverses the filters of one (sub-)image in place: rows of `stride` bytes, each preceded by its filter byte; returns the rows packed
std::vector<uint8_t> unfilter(const uint8_t *raw, size_t raw_len, int rows, size_t stride, int bpp)
{
    if (raw_len < (stride + 1) * (size_t)rows) bad("Corrupt PNG: not enough pixels");
    std::vector<uint8_t> img(stride * (size_t)rows);
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < rows; ++y) {
        const uint8_t *in = raw + (stride + 1) * (size_t)y;
        const int f = *in++;
        if (f > 4) bad("Corrupt PNG: invalid filter");
        uint8_t *cur = img.data() + stride * (size_t)y;
        const uint8_t *up = y ? cur - stride : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up[i], c = i >= (size_t)bpp ? up[i - bpp] : 0;
            int v = in[i];
            switch (f) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: break;
            }
            cur[i] = (uint8_t)v;
        }
    }
    return img;
}

ImageU8 decode_png_bytes(const uint8_t *data, size_t n)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (n < 8 || memcmp(data, sig, 8) != 0) bad("Corrupt PNG: bad png sig");
    const uint8_t *p = data + 8, *end = data + n;
    auto be32 = [&](const uint8_t *q) { return ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; };
    uint32_t W = 0, H = 0;
    int depth = 0, color = 0, interlace = 0;
    bool first = true, iphone = false, done = false;
    uint8_t palette[256][3];
    memset(palette, 0, sizeof(palette));
    int pal_len = 0;
    std::vector<uint8_t> idat;
    while (!done) {
        if (end - p < 8) bad("Corrupt PNG: truncated");
        const uint32_t len = be32(p), type = be32(p + 4);
        p += 8;
        if ((size_t)(end - p) < (size_t)len) bad("Corrupt PNG: outofdata");
        auto is = [&](const char *t) { return type == (((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | (uint32_t)t[3]); };
        if (is("CgBI")) {
            iphone = true;
        } else if (is("IHDR")) {
            if (!first) bad("Corrupt PNG: multiple IHDR");
            first = false;
            if (len != 13) bad("Corrupt PNG: bad IHDR len");
            W = be32(p); H = be32(p + 4);
            depth = p[8]; color = p[9];
            if (W > (1u << 24) || H > (1u << 24)) bad("Very large image (corrupt?)");
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) bad("PNG not supported: 1/2/4/8/16-bit only");
            if (color > 6 || (color == 3 && depth == 16) || (color != 3 && (color & 1))) bad("Corrupt PNG: bad ctype");
            if (p[10]) bad("Corrupt PNG: bad comp method");
            if (p[11]) bad("Corrupt PNG: bad filter method");
            interlace = p[12];
            if (interlace > 1) bad("Corrupt PNG: bad interlace method");
            if (!W || !H) bad("Corrupt PNG: 0-pixel image");
            if ((1u << 30) / W / 4 < H) bad("Image too large to decode");
        } else if (first) {
            bad("Corrupt PNG: first not IHDR");
        } else if (is("PLTE")) {
            if (len > 256 * 3 || len % 3) bad("Corrupt PNG: invalid PLTE");
            pal_len = (int)len / 3;
            memcpy(palette, p, len);
        } else if (is("IDAT")) {
            if (color == 3 && !pal_len) bad("Corrupt PNG: no PLTE");
            idat.insert(idat.end(), p, p + len);
        } else if (is("IEND")) {
            done = true;
        } else if (is("tRNS")) {
            // transparency only produces an alpha channel, which a 3-channel load drops
        } else if ((type & (1u << 29)) == 0) {
            bad("PNG not supported: unknown PNG chunk type");
        }
        p += len;
        if (!done) p += 4;     // CRC (not checked, like the reference's loader)
    }
    if (idat.empty()) bad("Corrupt PNG: no IDAT");
    Inflate z;
    z.p = idat.data(); z.end = idat.data() + idat.size();
    // the most bytes any layout of this image can hold: every Adam7 pass has at most H rows of at most the full row + filter byte
    z.limit = ((size_t)W * depth * 4 / 8 + 2) * ((size_t)H + 7) * (interlace ? 2 : 1);
    z.out.reserve(std::min<size_t>(z.limit, (size_t)64 << 20));
    z.run(!iphone);

    const int chans = color == 3 ? 1 : ((color & 2 ? 3 : 1) + (color & 4 ? 1 : 0));
    const int bpp = depth == 16 ? chans * 2 : chans;      // filter unit in bytes (1 for sub-byte depths of 1-channel images)
    // samples of the whole image as one byte each (16-bit: the high byte; sub-byte: the raw value)
    std::vector<uint8_t> smp((size_t)W * H * chans);
    auto unpack = [&](const std::vector<uint8_t> &img, uint32_t w, uint32_t h, size_t stride, uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
        for (uint32_t y = 0; y < h; ++y) {
            const uint8_t *row = img.data() + stride * y;
            for (uint32_t x = 0; x < w; ++x) {
                uint8_t *dst = smp.data() + ((size_t)(y0 + y * dy) * W + (x0 + x * dx)) * chans;
                for (int c = 0; c < chans; ++c) {
                    const size_t s = (size_t)x * chans + c;
                    if (depth == 8) dst[c] = row[s];
                    else if (depth == 16) dst[c] = row[2 * s];
                    else dst[c] = (uint8_t)((row[s * depth / 8] >> (8 - depth - (int)(s * depth % 8))) & ((1 << depth) - 1));
                }
            }
        }
    };
    if (!interlace) {
        const size_t stride = ((size_t)W * chans * depth + 7) / 8;
        unpack(unfilter(z.out.data(), z.out.size(), (int)H, stride, bpp), W, H, stride, 0, 0, 1, 1);
    } else {
        static const uint32_t xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t off = 0;
        for (int pass = 0; pass < 7; ++pass) {
            const uint32_t w = (W - xo[pass] + xs[pass] - 1) / xs[pass], h = (H - yo[pass] + ys[pass] - 1) / ys[pass];
            if (!w || !h) continue;
            const size_t stride = ((size_t)w * chans * depth + 7) / 8, need = (stride + 1) * h;
            if (off + need > z.out.size()) bad("Corrupt PNG: not enough pixels");
            unpack(unfilter(z.out.data() + off, need, (int)h, stride, bpp), w, h, stride, xo[pass], yo[pass], xs[pass], ys[pass]);
            off += need;
        }
    }
    ImageU8 im;
    im.w = (int)W; im.h = (int)H;
    im.rgb.resize((size_t)W * H * 3);
    static const int scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const uint8_t *s = smp.data() + i * chans;
        uint8_t *o = im.rgb.data() + i * 3;
        if (color == 3) { o[0] = palette[s[0]][0]; o[1] = palette[s[0]][1]; o[2] = palette[s[0]][2]; }
        else if (color & 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
        else o[0] = o[1] = o[2] = (uint8_t)(depth < 8 ? s[0] * scale[depth] : s[0]);
    }
    return im;
}

std::vector<uint8_t> read_file(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) bad("Cannot load image \"" + path + "\"");
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + k);
    fclose(f);
    return d;
}

}  // namespace

ImageU8 decode_jpeg(const uint8_t *data, size_t n)
{
    Jpeg j;
    j.p = data; j.end = data + n;
    j.decode();
    return jpeg_to_rgb(j);
}

ImageU8 decode_png(const uint8_t *data, size_t n) { return decode_png_bytes(data, n); }

ImageU8 decode_image(const uint8_t *data, size_t n, const std::string &name)
{
    try {
        if (n >= 2 && data[0] == 0xff && data[1] == 0xd8) return decode_jpeg(data, n);
        if (n >= 8 && data[0] == 137 && data[1] == 'P' && data[2] == 'N' && data[3] == 'G') return decode_png(data, n);
    } catch (const std::runtime_error &e) {
        throw std::runtime_error("Cannot load image \"" + name + "\"\nReason: " + e.what());
    }
    throw std::runtime_error("Cannot load image \"" + name + "\"\nReason: unknown image type (JPEG, PNG and binary PPM / PGM are supported)");
}

ImageU8 load_image_u8(const std::string &path)
{
    const std::vector<uint8_t> d = read_file(path);
    if (d.size() >= 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) return load_pnm_u8(path);
    return decode_image(d.data(), d.size(), path);
}

// load_image_stb (src/core/yolo_image.cpp:167-189) with channels = 3: HWC bytes -> CHW floats, (float)byte / 255.
Image load_image(const std::string &path)
{
    const ImageU8 u = load_image_u8(path);
    Image im = make_image(u.w, u.h, 3);
    for (int k = 0; k < 3; ++k)
        for (int y = 0; y < u.h; ++y)
            for (int x = 0; x < u.w; ++x) im.at(x, y, k) = (float)u.rgb[((size_t)y * u.w + x) * 3 + k] / 255.f;
    return im;
}

}  // namespace y2h
