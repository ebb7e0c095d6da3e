// jpeg_gray.hpp -- JPEG decoder (baseline, extended sequential and progressive Huffman) for single-component (grayscale) images.
//
// The reference loads its input with stb_image v2.26, `stbi_load(name, &w, &h, &n, 1)`
// (Deff2DGPU/Deff2D.cuh:342, cuh:377), and requires n == 1 (cuh:1665, cuh:1890).  Phase
// thresholds are applied to the decoded bytes, so the decoder's rounding matters: SURVEY.md
// section 5 records that libjpeg differs from stb_image by +-1 on thousands of pixels of the
// reference's 00042.jpg and flips several hundred across the 150 threshold.  stb_image.h is
// not redistributed here; this is an independent decoder for the files the reference accepts
// (baseline / extended-sequential / progressive Huffman, 8-bit, one component) that follows the same
// arithmetic so that it produces the same bytes:
//   * coefficients are dequantised into 16-bit storage (product truncated to int16),
//   * the inverse DCT is the LL&M "islow" scheme of the IJG library (jidctint) in 12-bit
//     fixed point: column pass keeps 2 extra bits (+512, >>10), row pass removes 17 bits with
//     rounding (+65536) and folds in the +128 level shift, then clamps to 0..255.
//   * progressive files (SOF2, ITU T.81 annex G: spectral selection and successive approximation) accumulate their
//     coefficients over the scans in the same 16-bit storage -- point transforms as shifts truncated to int16 -- and are
//     dequantised (product truncated to int16) and inverse-transformed once, after the last scan, like stb_image.h:3006-3025.
// Multi-component files are recognised (ncomp is reported) but not decoded: the reference
// rejects them too.  Arithmetic-coded and lossless files are rejected with a message, and so are PNG / BMP files, which
// the reference's stbi_load would read (stb_image.h:1094-1097): convert those to JPEG.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace deff {
namespace jpeg {

struct Huffman {
    // canonical code book, JPEG spec (ITU T.81) annex C / F.2.2.3
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
    void build(const uint8_t counts[16], const uint8_t *symbols, int nsym)
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memcpy(vals, symbols, (size_t)nsym);
        present = true;
    }
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill()
    {
        while (nbits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    int next = (p < end) ? *p : 0xD9;
                    if (next == 0x00) ++p;                         // stuffed zero
                    else { hit_marker = true; --p; byte = 0; }     // a marker: feed zeros from here on
                }
            }
            acc |= (uint32_t)byte << (24 - nbits);
            nbits += 8;
        }
    }
    int bit()
    {
        if (nbits < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1; --nbits;
        return b;
    }
    int bits(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n; nbits -= n;
        return v;
    }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader &br, const Huffman &h)
{
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
            return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
}

inline int extend(int v, int s)                                    // T.81 F.2.2.1
{
    return (v < (1 << (s - 1))) ? v - (1 << s) + 1 : v;
}

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// 12-bit fixed-point constants of the LL&M factorisation
constexpr int fx(double v) { return (int)(v * 4096 + 0.5); }
// (64-bit intermediates: the values of a valid file fit 32 bits with room to spare, so the
// results are those of 32-bit arithmetic; a damaged file can no longer overflow anything)
typedef int64_t idct_t;
struct Idct1D { idct_t e0, e1, e2, e3, o0, o1, o2, o3; };          // even part x0..x3, odd part t0..t3

inline Idct1D idct_1d(idct_t s0, idct_t s1, idct_t s2, idct_t s3, idct_t s4, idct_t s5, idct_t s6, idct_t s7)
{
    Idct1D r;
    // even part: rotation of (s2, s6), butterflies with (s0 +- s4) << 12
    idct_t z = (s2 + s6) * fx(0.5411961f);
    idct_t ev2 = z + s6 * fx(-1.847759065f);
    idct_t ev3 = z + s2 * fx(0.765366865f);
    idct_t ev0 = (s0 + s4) * 4096;
    idct_t ev1 = (s0 - s4) * 4096;
    r.e0 = ev0 + ev3; r.e3 = ev0 - ev3;
    r.e1 = ev1 + ev2; r.e2 = ev1 - ev2;
    // odd part
    idct_t a = s7, b = s5, c = s3, d = s1;
    idct_t ac = a + c, bd = b + d, ad = a + d, bc = b + c;
    idct_t z5 = (ac + bd) * fx(1.175875602f);
    a = a * fx(0.298631336f);
    b = b * fx(2.053119869f);
    c = c * fx(3.072711026f);
    d = d * fx(1.501321110f);
    ad = z5 + ad * fx(-0.899976223f);
    bc = z5 + bc * fx(-2.562915447f);
    ac = ac * fx(-1.961570560f);
    bd = bd * fx(-0.390180644f);
    r.o3 = d + ad + bd;
    r.o2 = c + bc + ac;
    r.o1 = b + bc + bd;
    r.o0 = a + ad + ac;
    return r;
}

inline uint8_t clamp255(idct_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

inline void idct_block(const int16_t c[64], uint8_t *out, int stride)
{
    idct_t tmp[64];
    for (int i = 0; i < 8; ++i) {                                  // columns, keep 2 extra bits
        Idct1D r = idct_1d(c[i], c[8 + i], c[16 + i], c[24 + i], c[32 + i], c[40 + i], c[48 + i], c[56 + i]);
        const idct_t rnd = 512;
        tmp[i] = (r.e0 + rnd + r.o3) >> 10;       tmp[56 + i] = (r.e0 + rnd - r.o3) >> 10;
        tmp[8 + i] = (r.e1 + rnd + r.o2) >> 10;   tmp[48 + i] = (r.e1 + rnd - r.o2) >> 10;
        tmp[16 + i] = (r.e2 + rnd + r.o1) >> 10;  tmp[40 + i] = (r.e2 + rnd - r.o1) >> 10;
        tmp[24 + i] = (r.e3 + rnd + r.o0) >> 10;  tmp[32 + i] = (r.e3 + rnd - r.o0) >> 10;
    }
    for (int i = 0; i < 8; ++i) {                                  // rows: 12 + 2 + 3 bits to remove
        const idct_t *v = tmp + 8 * i;
        Idct1D r = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const idct_t rnd = 65536 + (128 << 17);                       // rounding + level shift
        uint8_t *o = out + (size_t)i * stride;
        o[0] = clamp255((r.e0 + rnd + r.o3) >> 17);  o[7] = clamp255((r.e0 + rnd - r.o3) >> 17);
        o[1] = clamp255((r.e1 + rnd + r.o2) >> 17);  o[6] = clamp255((r.e1 + rnd - r.o2) >> 17);
        o[2] = clamp255((r.e2 + rnd + r.o1) >> 17);  o[5] = clamp255((r.e2 + rnd - r.o1) >> 17);
        o[3] = clamp255((r.e3 + rnd + r.o0) >> 17);  o[4] = clamp255((r.e3 + rnd - r.o0) >> 17);
    }
}

// One scan of a progressive one-component image over all its blocks (T.81 G.1.2): DC first / refinement (Ss = 0) or AC
// first / refinement (Ss > 0) with end-of-band runs.  `coef` holds 64 int16 per block, natural (de-zigzagged) order.
inline bool progressive_scan(BitReader &br, int16_t *coef, size_t nblocks, const Huffman *hdc, const Huffman *hac, int Ss, int Se,
                             int Ah, int Al, int restart, std::string &err)
{
    int64_t pred = 0;
    int eobrun = 0, todo = restart;
    const int16_t bit = (int16_t)(1 << Al);
    // successive-approximation correction of an already non-zero coefficient (G.1.2.3)
    auto refine = [&](int16_t &c) {
        if (br.bit() && (c & bit) == 0) c = (int16_t)(c > 0 ? c + bit : c - bit);
    };
    for (size_t blk = 0; blk < nblocks; ++blk) {
        int16_t *d = coef + blk * 64;
        if (Ss == 0) {
            if (Ah == 0) {
                const int t = decode_symbol(br, *hdc);
                if (t < 0 || t > 15) { err = "corrupt DC code"; return false; }
                pred += t ? extend(br.bits(t), t) : 0;
                d[0] = (int16_t)(pred * (1 << Al));
            } else if (br.bit()) {
                d[0] = (int16_t)(d[0] + bit);
            }
        } else if (Ah == 0) {
            if (eobrun) {
                --eobrun;
            } else {
                for (int k = Ss; k <= Se;) {
                    const int rs = decode_symbol(br, *hac);
                    if (rs < 0) { err = "corrupt AC code"; return false; }
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                        if (r < 15) {                              // end of band for this block and eobrun more
                            eobrun = (1 << r) - 1;
                            if (r) eobrun += br.bits(r);
                            break;
                        }
                        k += 16;
                    } else {
                        k += r;
                        if (k > 63) { err = "corrupt block"; return false; }
                        d[ZIGZAG[k++]] = (int16_t)(extend(br.bits(sz), sz) * (1 << Al));
                    }
                }
            }
        } else {
            if (eobrun) {
                --eobrun;
                for (int k = Ss; k <= Se; ++k)
                    if (d[ZIGZAG[k]] != 0) refine(d[ZIGZAG[k]]);
            } else {
                for (int k = Ss; k <= Se;) {
                    const int rs = decode_symbol(br, *hac);
                    if (rs < 0) { err = "corrupt AC code"; return false; }
                    int r = rs >> 4;
                    const int sz = rs & 15;
                    int16_t val = 0;
                    if (sz == 0) {
                        if (r < 15) {
                            eobrun = (1 << r) - 1;
                            if (r) eobrun += br.bits(r);
                            r = 64;                                // refine what is left of the band, place nothing
                        }
                    } else {
                        if (sz != 1) { err = "corrupt AC refinement code"; return false; }
                        val = br.bit() ? bit : (int16_t)-bit;
                    }
                    while (k <= Se) {                              // skip r zero coefficients, refining the non-zero ones passed
                        int16_t &c = d[ZIGZAG[k++]];
                        if (c != 0) {
                            refine(c);
                        } else {
                            if (r == 0) { c = val; break; }
                            --r;
                        }
                    }
                }
            }
        }
        if (restart && --todo == 0 && blk + 1 < nblocks) {
            // byte-align, expect RSTn, reset the predictor and the end-of-band run
            while (br.p < br.end && !(br.p[0] == 0xFF && br.p + 1 < br.end && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
            if (br.p + 2 <= br.end) br.p += 2;
            br.reset();
            pred = 0;
            eobrun = 0;
            todo = restart;
        }
    }
    return true;
}

// Returns true on success.  ncomp is set as soon as the frame header is seen, so a caller can
// report "n channels" for files that are not grayscale (the reference's check, cuh:1665).
inline bool decode_gray_impl(const uint8_t *data, size_t len, std::vector<uint8_t> &pix, int &w, int &h, int &ncomp,
                             std::string &err);

// the largest image accepted: 2^28 pixels (16384^2, the largest mesh of the benchmark configurations).  A progressive file
// keeps 128 B of coefficients per 64-pixel block over all its scans, so the bound also caps what a small crafted file can
// make the decoder allocate (2 B per pixel); an allocation failure is a decode error, not an exception.
constexpr uint64_t MAX_PIXELS = (uint64_t)1 << 28;

inline bool decode_gray(const uint8_t *data, size_t len, std::vector<uint8_t> &pix, int &w, int &h, int &ncomp,
                        std::string &err)
{
    try {
        return decode_gray_impl(data, len, pix, w, h, ncomp, err);
    } catch (const std::bad_alloc &) {
        err = "out of memory while decoding";
        return false;
    }
}

inline bool decode_gray_impl(const uint8_t *data, size_t len, std::vector<uint8_t> &pix, int &w, int &h, int &ncomp,
                             std::string &err)
{
    w = h = ncomp = 0;
    if (len >= 8 && !memcmp(data, "\x89PNG\r\n\x1a\n", 8)) {
        err = "PNG input: the reference's stb_image reads PNG too, this front end reads JPEG only -- convert the image to a one-component JPEG";
        return false;
    }
    if (len >= 2 && data[0] == 'B' && data[1] == 'M') {
        err = "BMP input: the reference's stb_image reads BMP too, this front end reads JPEG only -- convert the image to a one-component JPEG";
        return false;
    }
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) { err = "not a JPEG file (no SOI)"; return false; }
    bool progressive = false;
    int scans = 0;
    std::vector<int16_t> pcoef;                                    // progressive: 64 coefficients per block, all scans
    uint16_t quant[4][64];
    bool have_q[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    int restart = 0, qid = 0;
    size_t i = 2;
    while (i + 2 <= len) {
        if (data[i] != 0xFF) { err = "marker expected"; return false; }
        while (i < len && data[i] == 0xFF) ++i;                    // fill bytes
        if (i >= len) break;
        const int m = data[i++];
        if (m == 0xD9) {
            if (!progressive || !scans) { err = "EOI before any scan"; return false; }
            // progressive: all scans are in -- dequantise (int16 product, like every coefficient of the sequential path) and
            // inverse-transform every block
            if (!have_q[qid]) { err = "frame refers to a missing quantisation table"; return false; }
            const int bw = (w + 7) / 8, bh = (h + 7) / 8;
            std::vector<uint8_t> padded((size_t)bw * 8 * bh * 8);
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) {
                    int16_t *d = &pcoef[((size_t)by * bw + bx) * 64];
                    for (int z = 0; z < 64; ++z) d[z] = (int16_t)((int)d[z] * quant[qid][z]);
                    idct_block(d, padded.data() + ((size_t)by * 8 * bw + bx) * 8, bw * 8);
                }
            pix.resize((size_t)w * h);
            for (int y = 0; y < h; ++y) memcpy(&pix[(size_t)y * w], &padded[(size_t)y * bw * 8], (size_t)w);
            return true;
        }
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;       // TEM / stray RSTn: no payload
        if (i + 2 > len) break;
        const size_t L = ((size_t)data[i] << 8) | data[i + 1];
        if (L < 2 || i + L > len) { err = "truncated segment"; return false; }
        const uint8_t *seg = data + i + 2;
        const size_t n = L - 2;
        if (m == 0xDB) {                                           // quantisation tables
            size_t k = 0;
            while (k < n) {
                const int pq = seg[k] >> 4, tq = seg[k] & 15;
                ++k;
                if (tq > 3 || k + (pq ? 128u : 64u) > n) { err = "bad DQT"; return false; }
                for (int z = 0; z < 64; ++z) {
                    quant[tq][ZIGZAG[z]] = pq ? (uint16_t)((seg[k] << 8) | seg[k + 1]) : seg[k];
                    k += pq ? 2 : 1;
                }
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {                                    // Huffman tables
            size_t k = 0;
            while (k + 17 <= n) {
                const int tc = seg[k] >> 4, th = seg[k] & 15;
                int total = 0;
                for (int q = 0; q < 16; ++q) total += seg[k + 1 + q];
                if (th > 3 || tc > 1 || total > 256 || k + 17 + (size_t)total > n) { err = "bad DHT"; return false; }
                (tc ? ac[th] : dc[th]).build(seg + k + 1, seg + k + 17, total);
                k += 17 + (size_t)total;
            }
        } else if (m == 0xDD) {
            if (n < 2) { err = "bad DRI"; return false; }
            restart = (seg[0] << 8) | seg[1];
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {          // baseline / extended sequential / progressive
            if (n < 6) { err = "bad SOF"; return false; }
            if (w) { err = "second frame header"; return false; }
            progressive = m == 0xC2;
            if (seg[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
            h = (seg[1] << 8) | seg[2];
            w = (seg[3] << 8) | seg[4];
            ncomp = seg[5];
            if (w <= 0 || h <= 0) { err = "zero-sized image"; return false; }
            if ((uint64_t)w * (uint64_t)h > MAX_PIXELS) { err = "image larger than 2^28 pixels"; return false; }
            if (ncomp != 1) { err = "not a single-channel (grayscale) JPEG"; return false; }
            if (n < 9) { err = "bad SOF"; return false; }
            qid = seg[8] & 3;
        } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            if (n >= 6) ncomp = seg[5];
            err = "lossless / hierarchical / arithmetic-coded JPEG is not supported (Huffman baseline, extended and progressive only)";
            return false;
        } else if (m == 0xDA) {                                    // start of scan
            if (!w) { err = "SOS before SOF"; return false; }
            if (n < 6 || seg[0] != 1) { err = "bad SOS for a one-component image"; return false; }
            if (progressive) {
                const int td = seg[2] >> 4, ta = seg[2] & 15, Ss = seg[3], Se = seg[4], Ah = seg[5] >> 4, Al = seg[5] & 15;
                if (td > 3 || ta > 3 || Ss > Se || Se > 63 || Ah > 13 || Al > 13) { err = "bad progressive SOS"; return false; }
                if (Ss == 0 && Se != 0) { err = "progressive scan mixes DC and AC coefficients"; return false; }
                if (Ss == 0 ? (Ah == 0 && !dc[td].present) : !ac[ta].present) { err = "scan refers to a missing table"; return false; }
                const size_t bw = (size_t)(w + 7) / 8, bh = (size_t)(h + 7) / 8;
                if (pcoef.empty()) {
                    // the first (DC) scan spends at least one bit per block: a header that promises more blocks than the
                    // rest of the file can hold is damaged -- refuse before allocating
                    if ((uint64_t)bw * bh > (uint64_t)(len - (i + L)) * 8 + 1) { err = "scan data too short for the image size"; return false; }
                    pcoef.assign(bw * bh * 64, 0);
                }
                BitReader br(data + i + L, data + len);
                if (!progressive_scan(br, pcoef.data(), bw * bh, &dc[td], &ac[ta], Ss, Se, Ah, Al, restart, err)) return false;
                ++scans;
                // on to the marker that ends this scan's entropy-coded data (the reader never passes one)
                size_t pos = (size_t)(br.p - data);
                while (pos + 1 < len && !(data[pos] == 0xFF && data[pos + 1] != 0x00 && !(data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7))) ++pos;
                if (pos + 1 >= len) { err = "progressive JPEG ends without EOI"; return false; }
                i = pos;
                continue;
            }
            const int td = seg[2] >> 4, ta = seg[2] & 15;
            if (td > 3 || ta > 3 || !dc[td].present || !ac[ta].present || !have_q[qid]) { err = "scan refers to a missing table"; return false; }
            const int bw = (w + 7) / 8, bh = (h + 7) / 8;
            // a block takes at least 2 bits (shortest DC code + shortest end-of-block code): a header
            // that promises more blocks than the file can hold is damaged -- refuse before allocating
            if ((uint64_t)bw * bh > ((uint64_t)(len - (i + L)) * 8) / 2 + 1) { err = "scan data too short for the image size"; return false; }
            std::vector<uint8_t> padded((size_t)bw * 8 * bh * 8);
            BitReader br(data + i + L, data + len);
            int64_t pred = 0;
            int todo = restart;
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) {
                    int16_t coef[64];
                    memset(coef, 0, sizeof coef);
                    int t = decode_symbol(br, dc[td]);
                    if (t < 0 || t > 15) { err = "corrupt DC code"; return false; }
                    pred += t ? extend(br.bits(t), t) : 0;
                    coef[0] = (int16_t)(pred * quant[qid][0]);                 // int16 storage, as in stb_image
                    for (int k = 1; k < 64;) {
                        int rs = decode_symbol(br, ac[ta]);
                        if (rs < 0) { err = "corrupt AC code"; return false; }
                        const int r = rs >> 4, s = rs & 15;
                        if (s == 0) {
                            if (r != 15) break;                    // end of block
                            k += 16;
                        } else {
                            k += r;
                            if (k > 63) { err = "corrupt block"; return false; }
                            const int z = ZIGZAG[k++];
                            coef[z] = (int16_t)((int64_t)extend(br.bits(s), s) * quant[qid][z]);
                        }
                    }
                    idct_block(coef, padded.data() + ((size_t)by * 8 * bw + bx) * 8, bw * 8);
                    if (restart && --todo == 0 && !(by == bh - 1 && bx == bw - 1)) {
                        // byte-align, expect RSTn, reset the predictor
                        while (br.p < br.end && !(br.p[0] == 0xFF && br.p + 1 < br.end && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 2 <= br.end) br.p += 2;
                        br.reset();
                        pred = 0;
                        todo = restart;
                    }
                }
            pix.resize((size_t)w * h);
            for (int y = 0; y < h; ++y) memcpy(&pix[(size_t)y * w], &padded[(size_t)y * bw * 8], (size_t)w);
            return true;
        }
        i += L;
    }
    err = scans ? "progressive JPEG ends without EOI" : "no scan found";
    return false;
}

inline bool load_gray(const char *path, std::vector<uint8_t> &pix, int &w, int &h, int &ncomp, std::string &err)
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::vector<uint8_t> buf;
    uint8_t chunk[65536];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    fclose(f);
    return decode_gray(buf.data(), buf.size(), pix, w, h, ncomp, err);
}

}  // namespace jpeg
}  // namespace deff
