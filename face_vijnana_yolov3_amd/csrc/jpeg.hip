// JPEG decoding split between host and device (reference face_detection.py:112, 656, 798: `imread`).
//
// A baseline JPEG is a sequential Huffman bit stream (one symbol's length decides where the next begins: host work) in front
// of per-block arithmetic that is embarrassingly parallel (dequantise, 8x8 inverse DCT, chroma upsampling, YCbCr -> RGB:
// about half of libjpeg-turbo's time per image).  fv_jpeg_parse / fv_jpeg_entropy_decode run on host threads and emit the
// quantised coefficients (int16, natural order, one [64] block after another); fv_jpeg_reconstruct_batch turns the
// coefficients of a whole batch into packed RGB on the device, where fv_letterbox_batch picks them up -- the RGB image never
// exists on the host.  Arithmetic restated from the IJG / libjpeg-turbo sources so that the pixels are bit-identical to
// Pillow's (the reference's reader): jdhuff.c, jidctint.c jpeg_idct_islow, jdsample.c h2v1/h2v2_fancy_upsample, jdcolor.c
// ycc_rgb_convert; checked against Pillow itself (tests/test_jpeg_cpu.py, tests/test_jpeg_gpu.py) and against
// oracle/jpeg_oracle.py.  Supported: SOF0 / SOF1 (8-bit, Huffman), 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart
// intervals.  Everything else returns FV_ERR_INVALID and the caller decodes that file with Pillow.
#include <cstring>
#include "common.h"

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffSpec { uint8_t counts[16]; uint8_t symbols[256]; bool present; };

// derived decoding table: 9-bit look-ahead (code length <= 9: one lookup), maxcode / valoffset for longer codes
struct HuffDec {
    uint16_t look[512];          // (length << 8) | symbol, 0 = longer than 9 bits
    int32_t maxcode[18];
    int32_t valoff[17];
    uint8_t symbols[256];
};

bool build_table(const HuffSpec& s, HuffDec& d) {
    int code = 0, k = 0;
    memset(d.look, 0, sizeof d.look);
    memcpy(d.symbols, s.symbols, 256);
    for (int len = 1; len <= 16; ++len) {
        d.valoff[len] = k - code;
        for (int i = 0; i < s.counts[len - 1]; ++i, ++code, ++k) {
            if (k >= 256 || code >= (1 << len)) return false;
            if (len <= 9) {
                const int base = code << (9 - len);
                for (int f = 0; f < (1 << (9 - len)); ++f) d.look[base + f] = (uint16_t)((len << 8) | s.symbols[k]);
            }
        }
        d.maxcode[len] = s.counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    d.maxcode[17] = 0x7FFFFFFF;
    return true;
}

struct Parsed {
    fv_jpeg_info info;
    HuffSpec dc[4], ac[4];
    int td[3], ta[3];
    size_t scan_begin;
};

int parse(const uint8_t* b, size_t n, Parsed& P) {
    memset(&P, 0, sizeof P);
    if (n < 4 || b[0] != 0xFF || b[1] != 0xD8) return FV_ERR_INVALID;
    size_t p = 2;
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false}, have_sof = false;
    int tq[3] = {0, 0, 0}, cid[3] = {0, 0, 0};
    int adobe = -1;
    while (p + 4 <= n) {
        if (b[p] != 0xFF) return FV_ERR_INVALID;
        while (p + 1 < n && b[p + 1] == 0xFF) ++p;      // fill bytes
        if (p + 2 > n) return FV_ERR_INVALID;             // the buffer ended inside a run of fill bytes: no marker code left
        const int m = b[p + 1];
        p += 2;
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (p + 2 > n) return FV_ERR_INVALID;
        const size_t len = ((size_t)b[p] << 8) | b[p + 1];
        if (len < 2 || p + len > n) return FV_ERR_INVALID;
        const uint8_t* s = b + p + 2;
        const size_t sl = len - 2;
        if (m == 0xDB) {
            size_t q = 0;
            while (q + 65 <= sl) {
                const int pq = s[q] >> 4, t = s[q] & 15;
                if (pq || t > 3) return FV_ERR_INVALID;
                for (int i = 0; i < 64; ++i) qt[t][kZigzag[i]] = s[q + 1 + i];
                have_qt[t] = true;
                q += 65;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return FV_ERR_INVALID;
            P.info.height = (s[1] << 8) | s[2];
            P.info.width = (s[3] << 8) | s[4];
            P.info.ncomp = s[5];
            if ((P.info.ncomp != 1 && P.info.ncomp != 3) || sl < 6 + 3 * (size_t)P.info.ncomp) return FV_ERR_INVALID;
            for (int i = 0; i < P.info.ncomp; ++i) {
                cid[i] = s[6 + 3 * i];
                P.info.h[i] = s[7 + 3 * i] >> 4; P.info.v[i] = s[7 + 3 * i] & 15;
                tq[i] = s[8 + 3 * i];
                if (tq[i] > 3) return FV_ERR_INVALID;
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return FV_ERR_INVALID;                                   // progressive, lossless, arithmetic coding
        } else if (m == 0xC4) {
            size_t q = 0;
            while (q + 17 <= sl) {
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3) return FV_ERR_INVALID;
                HuffSpec& h = tc ? P.ac[th] : P.dc[th];
                int ns = 0;
                for (int i = 0; i < 16; ++i) { h.counts[i] = s[q + 1 + i]; ns += h.counts[i]; }
                if (ns > 256 || q + 17 + ns > sl) return FV_ERR_INVALID;
                memcpy(h.symbols, s + q + 17, ns);
                h.present = true;
                q += 17 + ns;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return FV_ERR_INVALID;
            P.info.restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xEE && sl >= 12 && memcmp(s, "Adobe", 5) == 0) {
            adobe = s[11];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1 || s[0] != P.info.ncomp || sl < 1 + 2 * (size_t)s[0]) return FV_ERR_INVALID;   // one interleaved scan
            for (int i = 0; i < P.info.ncomp; ++i) {
                int c = -1;
                for (int j = 0; j < P.info.ncomp; ++j) if (cid[j] == s[1 + 2 * i]) c = j;
                if (c != i) return FV_ERR_INVALID;
                P.td[i] = s[2 + 2 * i] >> 4; P.ta[i] = s[2 + 2 * i] & 15;
                if (P.td[i] > 3 || P.ta[i] > 3 || !P.dc[P.td[i]].present || !P.ac[P.ta[i]].present) return FV_ERR_INVALID;
            }
            P.scan_begin = p + len;
            break;
        }
        p += len;
    }
    if (!have_sof || !P.scan_begin || P.info.width < 1 || P.info.height < 1) return FV_ERR_INVALID;
    if (P.info.ncomp == 3 && adobe != -1 && adobe != 1) return FV_ERR_INVALID;              // Adobe RGB / CMYK-style files
    int hmax = 1, vmax = 1;
    for (int i = 0; i < P.info.ncomp; ++i) {
        if (P.info.h[i] < 1 || P.info.h[i] > 2 || P.info.v[i] < 1 || P.info.v[i] > 2 || !have_qt[tq[i]]) return FV_ERR_INVALID;
        hmax = P.info.h[i] > hmax ? P.info.h[i] : hmax; vmax = P.info.v[i] > vmax ? P.info.v[i] : vmax;
    }
    if (P.info.ncomp == 1) { P.info.h[0] = P.info.v[0] = 1; hmax = vmax = 1; }               // a single component is never subsampled
    else if (P.info.h[0] != hmax || P.info.v[0] != vmax || P.info.h[1] != 1 || P.info.v[1] != 1 || P.info.h[2] != 1 || P.info.v[2] != 1 ||
             (hmax == 1 && vmax == 2))
        return FV_ERR_INVALID;                                       // 4:4:4, 4:2:2 (2x1), 4:2:0 (2x2) only
    const int mcux = (P.info.width + 8 * hmax - 1) / (8 * hmax), mcuy = (P.info.height + 8 * vmax - 1) / (8 * vmax);
    int64_t off = 0;
    for (int i = 0; i < P.info.ncomp; ++i) {
        P.info.blocks_w[i] = mcux * P.info.h[i]; P.info.blocks_h[i] = mcuy * P.info.v[i];
        P.info.coef_off[i] = off;
        off += (int64_t)P.info.blocks_w[i] * P.info.blocks_h[i] * 64;
        for (int k = 0; k < 64; ++k) P.info.qt[i][k] = qt[tq[i]][k];
    }
    P.info.total_coefs = off;
    P.info.hmax = hmax; P.info.vmax = vmax;
    return FV_OK;
}

// ---- bit reader (byte stuffing: FF 00 -> FF; any other FF xx is a marker: stop in front of it and feed zeros)
struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int nbits = 0;
    bool marker = false;
    inline void fill() {
        while (nbits <= 56) {
            unsigned byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    if (p < end && *p == 0) ++p;
                    else { --p; marker = true; byte = 0; }
                }
            }
            acc = (acc << 8) | byte;
            nbits += 8;
        }
    }
    inline unsigned peek(int n) { return (unsigned)((acc >> (nbits - n)) & ((1u << n) - 1)); }
    inline void skip(int n) { nbits -= n; }
    inline unsigned get(int n) { unsigned v = peek(n); nbits -= n; return v; }
};

inline int extend(unsigned v, int n) { return (int)v < (1 << (n - 1)) ? (int)v - (1 << n) + 1 : (int)v; }

inline int decode_symbol(BitReader& br, const HuffDec& d) {
    if (br.nbits < 16) br.fill();
    const unsigned l = d.look[br.peek(9)];
    if (l) { br.skip(l >> 8); return l & 0xFF; }
    int len = 10;
    int code = (int)br.peek(10);
    while (code > d.maxcode[len]) { ++len; if (len > 16) return -1; code = (int)br.peek(len); }
    br.skip(len);
    return d.symbols[(code + d.valoff[len]) & 0xFF];
}

}  // namespace

extern "C" {

int fv_jpeg_parse(const uint8_t* data, size_t nbytes, fv_jpeg_info* info) {
    if (!data || !info) return FV_ERR_INVALID;
    Parsed P;
    const int rc = parse(data, nbytes, P);
    if (rc == FV_OK) *info = P.info;
    return rc;
}

int fv_jpeg_entropy_decode(const uint8_t* data, size_t nbytes, int16_t* coefs, int64_t ncoefs) {
    if (!data || !coefs) return FV_ERR_INVALID;
    Parsed P;
    if (int rc = parse(data, nbytes, P)) return rc;
    if (ncoefs < P.info.total_coefs) return FV_ERR_WORKSPACE;
    HuffDec dc[4], ac[4];
    for (int t = 0; t < 4; ++t) {
        if (P.dc[t].present && !build_table(P.dc[t], dc[t])) return FV_ERR_INVALID;
        if (P.ac[t].present && !build_table(P.ac[t], ac[t])) return FV_ERR_INVALID;
    }
    memset(coefs, 0, (size_t)P.info.total_coefs * sizeof(int16_t));
    const fv_jpeg_info& I = P.info;
    const int mcux = I.blocks_w[0] / I.h[0], mcuy = I.blocks_h[0] / I.v[0];
    BitReader br{data + P.scan_begin, data + nbytes};
    int pred[3] = {0, 0, 0};
    int to_go = I.restart_interval;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (I.restart_interval) {
                if (to_go == 0) {
                    // byte-align, step over the RSTn marker (resynchronise on the next one if the stream is damaged)
                    br.nbits = 0; br.acc = 0; br.marker = false;
                    const uint8_t* q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                    br.p = q + 2 <= br.end ? q + 2 : br.end;
                    pred[0] = pred[1] = pred[2] = 0;
                    to_go = I.restart_interval;
                }
                --to_go;
            }
            for (int c = 0; c < I.ncomp; ++c) {
                const HuffDec& hd = dc[P.td[c]];
                const HuffDec& ha = ac[P.ta[c]];
                for (int by = 0; by < I.v[c]; ++by)
                    for (int bx = 0; bx < I.h[c]; ++bx) {
                        int16_t* blk = coefs + I.coef_off[c] + ((int64_t)(my * I.v[c] + by) * I.blocks_w[c] + mx * I.h[c] + bx) * 64;
                        int s = decode_symbol(br, hd);
                        if (s < 0 || s > 11) return FV_ERR_INVALID;
                        if (s) {
                            if (br.nbits < s) br.fill();
                            pred[c] += extend(br.get(s), s);
                        }
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = decode_symbol(br, ha);
                            if (rs < 0) return FV_ERR_INVALID;
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) return FV_ERR_INVALID;
                            if (br.nbits < sz) br.fill();
                            blk[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
                            ++k;
                        }
                    }
            }
        }
    return FV_OK;
}

}  // extern "C"

// ====================================================================================================== device side
namespace {

__device__ __forceinline__ int descale(long long x, int n) { return (int)((x + (1ll << (n - 1))) >> n); }

// jidctint.c jpeg_idct_islow, one 1-D pass over eight values (CONST_BITS = 13; 64-bit temporaries like the JLONG of the C code)
__device__ __forceinline__ void idct_1d(const int (&v)[8], int (&o)[8], int shift) {
    constexpr long long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                        F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    long long z2 = v[2], z3 = v[6];
    long long z1 = (z2 + z3) * F0541;
    long long t2 = z1 + z3 * (-F1847), t3 = z1 + z2 * F0765;
    z2 = v[0]; z3 = v[4];
    long long t0 = (z2 + z3) << 13, t1 = (z2 - z3) << 13;
    const long long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    long long z4 = t1 + t3;
    const long long z5 = (z3 + z4) * F1175;
    t0 *= F0298; t1 *= F2053; t2 *= F3072; t3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 = z3 * (-F1961) + z5; z4 = z4 * (-F0390) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    o[0] = descale(t10 + t3, shift); o[7] = descale(t10 - t3, shift);
    o[1] = descale(t11 + t2, shift); o[6] = descale(t11 - t2, shift);
    o[2] = descale(t12 + t1, shift); o[5] = descale(t12 - t1, shift);
    o[3] = descale(t13 + t0, shift); o[4] = descale(t13 - t0, shift);
}

// one thread = one 8x8 block: dequantise, two passes, range-limit, store into the component plane (block grid padded to MCUs)
__global__ __launch_bounds__(128) void jpeg_idct_kernel(const int16_t* __restrict__ coefs, const fv_jpeg_desc* __restrict__ descs,
                                                        uint8_t* __restrict__ planes) {
    const fv_jpeg_desc& d = descs[blockIdx.y];
    __shared__ uint16_t qt[3][64];
    for (int i = threadIdx.x; i < 192; i += blockDim.x) qt[i / 64][i % 64] = d.qt[i / 64][i % 64];
    __syncthreads();
    const long long nb0 = (long long)d.blocks_w[0] * d.blocks_h[0];
    const long long nb1 = d.ncomp == 3 ? (long long)d.blocks_w[1] * d.blocks_h[1] : 0;
    const long long total = nb0 + 2 * nb1;
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        int c = 0;
        long long b = g;
        if (b >= nb0) { b -= nb0; c = 1; if (b >= nb1) { b -= nb1; c = 2; } }
        const int bw = d.blocks_w[c];
        const int by = (int)(b / bw), bx = (int)(b - (long long)by * bw);
        const int16_t* src = coefs + d.coef_off[c] + b * 64;
        int ws[8][8];
        // pass 1: columns
#pragma unroll
        for (int col = 0; col < 8; ++col) {
            int v[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (int)src[k * 8 + col] * (int)qt[c][k * 8 + col];
            idct_1d(v, o, 13 - 2);
#pragma unroll
            for (int k = 0; k < 8; ++k) ws[k][col] = o[k];
        }
        // pass 2: rows
        uint8_t* dst = planes + d.plane_off[c] + ((long long)by * 8) * (bw * 8) + bx * 8;
#pragma unroll
        for (int row = 0; row < 8; ++row) {
            int o[8];
            idct_1d(ws[row], o, 13 + 2 + 3);
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int a = o[k] + 128, e = o[4 + k] + 128;
                a = a < 0 ? 0 : (a > 255 ? 255 : a); e = e < 0 ? 0 : (e > 255 ? 255 : e);
                lo |= (uint32_t)a << (8 * k); hi |= (uint32_t)e << (8 * k);
            }
            *reinterpret_cast<uint2*>(dst + (long long)row * (bw * 8)) = make_uint2(lo, hi);
        }
    }
}

// chroma sample at full resolution (jdsample.c fancy upsampling; edges replicate the last real row / column, jdmainct.c)
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int stride, int cw, int ch, int x, int y, int fh, int fv) {
    if (fh == 1) return p[(long long)y * stride + x];
    const int cx = x >> 1;
    if (fv == 1) {                                                // h2v1
        const int v = p[(long long)y * stride + cx];
        if (cw == 1) return v;
        if (x & 1) return cx == cw - 1 ? v : (3 * v + p[(long long)y * stride + cx + 1] + 2) >> 2;
        return cx == 0 ? v : (3 * v + p[(long long)y * stride + cx - 1] + 1) >> 2;
    }
    // h2v2: the nearer vertical neighbour (above for even output rows, below for odd), replicated at the image edge
    const int cy = y >> 1;
    int ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : (ny > ch - 1 ? ch - 1 : ny);
    const uint8_t* r0 = p + (long long)cy * stride;
    const uint8_t* r1 = p + (long long)ny * stride;
    const int col = 3 * r0[cx] + r1[cx];
    if (x & 1) {
        if (cx == cw - 1) return (4 * col + 7) >> 4;
        return (3 * col + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
    }
    if (cx == 0) return (4 * col + 8) >> 4;
    return (3 * col + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
}

// one thread = one output pixel: upsample, convert (jdcolor.c ycc_rgb_convert), store packed RGB
__global__ __launch_bounds__(256) void jpeg_color_kernel(const fv_jpeg_desc* __restrict__ descs, const uint8_t* __restrict__ planes,
                                                         uint8_t* __restrict__ rgb) {
    const fv_jpeg_desc& d = descs[blockIdx.y];
    const long long npix = (long long)d.width * d.height;
    uint8_t* out = rgb + d.rgb_off;
    const uint8_t* py = planes + d.plane_off[0];
    const int sy = d.blocks_w[0] * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / d.width), x = (int)(i - (long long)y * d.width);
        const int Y = py[(long long)y * sy + x];
        int R = Y, G = Y, B = Y;
        if (d.ncomp == 3) {
            const int fh = d.hmax, fv = d.vmax;
            const int cw = (d.width + fh - 1) / fh, ch = (d.height + fv - 1) / fv;
            const int cb = chroma_at(planes + d.plane_off[1], d.blocks_w[1] * 8, cw, ch, x, y, fh, fv) - 128;
            const int cr = chroma_at(planes + d.plane_off[2], d.blocks_w[2] * 8, cw, ch, x, y, fh, fv) - 128;
            R = Y + ((91881 * cr + 32768) >> 16);
            B = Y + ((116130 * cb + 32768) >> 16);
            G = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
            R = R < 0 ? 0 : (R > 255 ? 255 : R); G = G < 0 ? 0 : (G > 255 ? 255 : G); B = B < 0 ? 0 : (B > 255 ? 255 : B);
        }
        out[3 * i] = (uint8_t)R; out[3 * i + 1] = (uint8_t)G; out[3 * i + 2] = (uint8_t)B;
    }
}

}  // namespace

extern "C" {

int64_t fv_jpeg_plane_bytes(const fv_jpeg_info* info) {
    if (!info) return 0;
    int64_t n = 0;
    for (int c = 0; c < info->ncomp; ++c) n += (int64_t)info->blocks_w[c] * info->blocks_h[c] * 64;
    return (n + 15) & ~(int64_t)15;
}

int fv_jpeg_reconstruct_batch(fv_ctx* ctx, const int16_t* coefs, const fv_jpeg_desc* descs, int n, uint8_t* planes, uint8_t* rgb,
                              int64_t max_blocks, int64_t max_pixels) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, coefs && descs && planes && rgb && n >= 1 && max_blocks >= 1 && max_pixels >= 1, "jpeg_reconstruct_batch: bad arguments");
    {
        FvProfScope ps(ctx, "jpeg_idct_kernel", 0.0, 0.0);
        long long gx = (max_blocks + 127) / 128;
        gx = gx > 2048 ? 2048 : gx;
        hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)gx, (unsigned)n), dim3(128), 0, ctx->stream, coefs, descs, planes);
        FV_LAUNCH_CHECK(ctx);
    }
    {
        FvProfScope ps(ctx, "jpeg_color_kernel", 0.0, 0.0);
        long long gx = (max_pixels + 255) / 256;
        gx = gx > 4096 ? 4096 : gx;
        hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, ctx->stream, descs, planes, rgb);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

}  // extern "C"
