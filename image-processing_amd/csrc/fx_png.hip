// libfx: PNG pixel data decoded on the device -- the inflate (RFC 1951 inside the zlib wrapper of RFC 1950) and the
// row filters of PNG (filter method 0), then PIL's convert("L").  The host has checked the container and staged
// each file's zlib stream (fx_png_stage); what comes out equals fx_decode_png_gray_u8 = PIL's
// Image.open(path).convert("L") (reference src/features/extract_2d_hu.py:84-98) byte for byte, or a status.
//
// A deflate stream is decoded one symbol after the other, so a file is one wave's work and within the wave one
// lane's: lane 0 walks the bit stream, the other lanes move the compressed bytes into an LDS ring ahead of it and
// fill the code tables.  Thousands of files at a time keep the chip busy.
#include "fx_common.h"

namespace {

constexpr int LIT_BITS = 10, DIST_BITS = 8;   // codes up to this length are looked up, longer ones walked bit by bit
constexpr int RING_WORDS = 512;               // compressed bytes ahead of the decoder, in LDS (2 KiB)
constexpr int MAX_LIT = 288, MAX_DIST = 32, MAX_LENS = MAX_LIT + MAX_DIST;
constexpr int PNG_MAX_ROW = 4096;             // W * channels

enum { PNG_OK = 0, PNG_EHEADER = 1, PNG_EBLOCK = 2, PNG_ECODES = 3, PNG_ESYMBOL = 4, PNG_EDISTANCE = 5, PNG_ELENGTH = 6,
       PNG_EADLER = 7, PNG_EFILTER = 8, PNG_EINPUT = 9 };

struct PngArgs {
    const uint8_t *staged;
    const int64_t *offset;
    const int32_t *length;
    const int32_t *channels;
    const int64_t *raw_offset;   // [n + 1] where a file's unfiltered bytes start in `raw`
    int64_t n;
    int H, W;
    int max_ch;                  // channels a row of W pixels may have (PNG_MAX_ROW bytes of LDS per row)
    uint8_t *raw;
    uint32_t *adler;             // [n] the Adler-32 a stream ends with (checked where the bytes are read next)
    uint8_t *out;
    int32_t *status;
};

struct __align__(16) InflateLds {
    uint32_t ring[RING_WORDS];
    uint16_t lit_fast[1 << LIT_BITS];    // (symbol << 4) | code length; 0: a longer code
    uint16_t dist_fast[1 << DIST_BITS];
    uint16_t lit_count[16], dist_count[16];   // codes of each length
    uint16_t lit_sym[MAX_LIT], dist_sym[MAX_DIST];   // symbols in order of (length, value)
    uint16_t code[MAX_LENS];             // canonical code of a symbol (while the tables are filled)
    uint8_t lens[MAX_LENS];              // code lengths of the block being set up
};

__device__ const uint16_t k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t k_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The bit stream as lane 0 sees it: bits come out of the low end of `buf`; `next` is the next ring word.
struct BitReader {
    u64 buf;
    int cnt;
    uint32_t next;
    const uint32_t *ring;
    __device__ __forceinline__ void fill()
    {
        if (cnt <= 32) {
            buf |= (u64)ring[next & (RING_WORDS - 1)] << cnt;
            cnt += 32;
            next++;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n)
    {
        buf >>= n;
        cnt -= n;
    }
    __device__ __forceinline__ uint32_t take(int n)   // n <= 16; filled beforehand
    {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

// A symbol of a code given by its fast table and, for the longer codes, its counts and ordered symbols (the walk of
// zlib's contrib/puff: the codes of one length are consecutive, first the shortest).  -1: no such code.
__device__ __forceinline__ int decode_symbol(BitReader &br, const uint16_t *fast, int fast_bits, const uint16_t *count,
                                             const uint16_t *sym)
{
    br.fill();
    const uint32_t e = fast[br.peek(fast_bits)];
    if (e) {
        br.drop((int)(e & 15u));
        return (int)(e >> 4);
    }
    int code = 0, first = 0, index = 0;
    u64 bits = br.buf;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1ull);
        bits >>= 1;
        const int c = count[len];
        if (code - c < first) {
            br.drop(len);
            return sym[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Tables of one code from the lengths lens[0 .. n): counts, ordered symbols, canonical codes (lane 0), then the fast
// table (all lanes).  Returns false for an over-subscribed set of lengths.  Every lane calls.
__device__ __forceinline__ bool build_code(const uint8_t *lens, int n, uint16_t *count, uint16_t *sym, uint16_t *code,
                                           uint16_t *fast, int fast_bits, bool strict)
{
    const int lane = fx_lane();
    int ok = 1;
    if (lane == 0) {
        for (int l = 0; l < 16; l++)
            count[l] = 0;
        for (int s = 0; s < n; s++)
            count[lens[s]]++;
        // over-subscribed: never; incomplete: only a code whose symbols all have one bit (zlib's rule), or the
        // fixed code's 30 distance symbols of five bits
        int left = 1, longest = 0, cc = 0;
        uint16_t offs[16], next_code[16];
        offs[1] = 0;
        for (int l = 1; l <= 15; l++) {
            left <<= 1;
            left -= count[l];
            if (left < 0)
                ok = 0;
            if (count[l])
                longest = l;
            if (l < 15)
                offs[l + 1] = (uint16_t)(offs[l] + count[l]);
            next_code[l] = (uint16_t)cc;
            cc = (cc + count[l]) << 1;
        }
        if (left > 0 && strict && longest != 1)
            ok = 0;
        if (ok) {
            for (int s = 0; s < n; s++) {
                const int l = lens[s];
                if (l) {
                    sym[offs[l]++] = (uint16_t)s;
                    code[s] = next_code[l]++;
                }
            }
        }
        count[0] = 0;   // (symbols without a code are not part of the walk)
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    for (int k = lane; k < (1 << fast_bits); k += FX_WAVE)
        fast[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!ok)
        return false;
    for (int s = lane; s < n; s += FX_WAVE) {
        const int l = lens[s];
        if (l && l <= fast_bits) {
            const uint32_t rev = __brev((uint32_t)code[s]) >> (32 - l);   // the stream carries a code's first bit first
            const uint16_t e = (uint16_t)((s << 4) | l);
            for (uint32_t k = rev; k < (1u << fast_bits); k += 1u << l)
                fast[k] = e;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return true;
}

constexpr int COPY_TOGETHER = 16;   // matches of at least this many bytes are copied by all lanes at once

// A wave per file.  Output: the file's H rows of (filter byte, W * channels bytes) in `raw`, the Adler-32 its stream
// closes with, its status.
__global__ __launch_bounds__(FX_WAVE) void png_inflate_kernel(PngArgs A)
{
    __shared__ InflateLds L;
    const int lane = fx_lane();
    const int64_t f = blockIdx.x;
    const uint32_t *z = reinterpret_cast<const uint32_t *>(A.staged + A.offset[f]);
    const int ch_f = A.channels[f];
    // (else: a header status, and not a byte of the stream is read.  Deflate grows incompressible data by a few bytes
    // per 64 KiB; a stream of more than twice the bytes it is to produce is left to the host decoder, which also
    // bounds the time a wave can spend on bytes made up to keep it busy.)
    const int64_t raw_bytes = A.raw_offset[f + 1] - A.raw_offset[f];
    const bool listed = A.length[f] >= 6 && ch_f >= 1 && ch_f <= A.max_ch && (int64_t)A.length[f] <= 2 * raw_bytes + 1024;
    const int zlen = listed ? A.length[f] : 0;
    const uint32_t zwords = listed ? (uint32_t)((zlen + 3) >> 2) + 4u : 0u;   // (the host left 16 zero bytes behind the stream)
    uint8_t *out = A.raw + A.raw_offset[f];   // (written and, for matches, read back by lane 0 alone)
    const int64_t want = A.raw_offset[f + 1] - A.raw_offset[f];

    // ring: words [filled - RING_WORDS, filled) of the stream
    uint32_t filled = 0;
    const auto refill = [&](uint32_t upto) {   // every lane; `upto` wave-uniform
        while (filled < upto) {
            const uint32_t w = filled + (uint32_t)lane;
            if (w < upto)
                L.ring[w & (RING_WORDS - 1)] = w < zwords ? z[w] : 0u;
            filled += FX_WAVE;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    refill(RING_WORDS);

    BitReader br;
    br.buf = 0;
    br.cnt = 0;
    br.next = 0;
    br.ring = L.ring;
    int64_t op = 0;            // bytes written
    int status = PNG_OK;
    int copy_len = 0, copy_dist = 0;   // a long match that lane 0 has met: the wave copies it at the top of the loop
    int phase = 0;             // 0: at a block header, 1: in a coded block, 2: in a stored block, 3: after the last block
    int last = 0;              // the block in hand is the final one
    int stored_left = 0;
    int setup = 0;             // a coded block's lengths are in L.lens: tables to be built (1: fixed, 2: dynamic)
    int n_lit = 0, n_dist = 0;

    if (lane == 0) {
        // zlib wrapper: deflate, window <= 32 KiB, no preset dictionary, header check
        br.fill();
        const uint32_t cmf = br.take(8), flg = br.take(8);
        if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 32u) || ((cmf << 8) | flg) % 31u != 0u || !listed)
            status = PNG_EHEADER;
    }

    // (Every round of this loop uses up bits of the stream, or follows one that did -- the round that builds a
    // block's tables.  A long match takes a round of its own and may cost two bits, one for a length code and one
    // for a distance code that are each the only one of their kind: a flat image of a megabyte is a few thousand
    // such matches in a few hundred bytes.  Two rounds per bit is a bound no stream that decodes can reach, and it
    // ends the wave whatever the bytes are.)
    const long long max_rounds = 16ll * (long long)zlen + 64;
    for (long long round = 0;; round++) {
        // ---- every lane: state from lane 0, the ring kept at least half full ahead of the reader
        if (round > max_rounds)
            status = PNG_EINPUT;
        status = __builtin_amdgcn_readfirstlane(status);
        phase = __builtin_amdgcn_readfirstlane(phase);
        setup = __builtin_amdgcn_readfirstlane(setup);
        copy_len = __builtin_amdgcn_readfirstlane(copy_len);
        if (status != PNG_OK)
            break;
        if (copy_len > 0) {
            // Byte k of the match is byte (k mod dist) of the dist bytes in front of it: all of them written already,
            // so the lanes copy side by side (one trip to memory for the match instead of one per byte).
            const int dist = __builtin_amdgcn_readfirstlane(copy_dist);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)op);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((u64)op >> 32));
            const int64_t at = (int64_t)(((u64)hi << 32) | lo);
            for (int k = lane; k < copy_len; k += FX_WAVE)
                out[at + k] = out[at - dist + (dist >= copy_len ? k : k % dist)];
            op = at + copy_len;
            copy_len = 0;
        }
        if (phase == 3)
            break;
        {
            const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)br.next);
            // words [next, next + RING_WORDS / 2 + ...) must be in the ring; none of [next - 2, next) may be overwritten
            const uint32_t upto = (next & ~(uint32_t)(FX_WAVE - 1)) + RING_WORDS - FX_WAVE;
            if (filled < upto)
                refill(upto);
        }
        if (setup) {
            n_lit = __builtin_amdgcn_readfirstlane(n_lit);
            n_dist = __builtin_amdgcn_readfirstlane(n_dist);
            const bool ok1 = build_code(L.lens, n_lit, L.lit_count, L.lit_sym, L.code, L.lit_fast, LIT_BITS, setup == 2);
            const bool ok2 = build_code(L.lens + n_lit, n_dist, L.dist_count, L.dist_sym, L.code + MAX_LIT, L.dist_fast, DIST_BITS,
                                        setup == 2);
            if (!ok1 || !ok2)
                status = PNG_ECODES;
            setup = 0;
            phase = 1;
            continue;
        }
        if (lane != 0)
            continue;   // (lane 0 alone from here to the end of the round; the others meet it at the top)

        // the reader may use up to this word before the ring is looked after again
        const uint32_t limit = filled - 8u;

        if (phase == 0) {
            // ---- block header (a dynamic header is at most 14 + 19 * 3 + 320 * 14 bits: well inside the ring)
            br.fill();
            last = (int)br.take(1);
            const uint32_t type = br.take(2);
            if (type == 0u) {
                br.drop(br.cnt & 7);   // to the byte boundary
                br.fill();
                const uint32_t len = br.take(16);
                br.fill();
                const uint32_t nlen = br.take(16);
                if ((len ^ nlen) != 0xffffu)
                    status = PNG_EBLOCK;
                stored_left = (int)len;
                phase = 2;
            } else if (type == 1u) {
                for (int s = 0; s < 288; s++)
                    L.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                for (int s = 0; s < 30; s++)
                    L.lens[288 + s] = 5;
                n_lit = 288;
                n_dist = 30;
                setup = 1;
            } else if (type == 2u) {
                br.fill();
                n_lit = (int)br.take(5) + 257;
                n_dist = (int)br.take(5) + 1;
                const int n_clen = (int)br.take(4) + 4;
                if (n_lit > 286 || n_dist > 30)
                    status = PNG_EBLOCK;
                // the code of the code lengths: 19 symbols of up to 7 bits, walked bit by bit
                uint8_t cl[19];
                for (int k = 0; k < 19; k++)
                    cl[k] = 0;
                for (int k = 0; k < n_clen; k++) {
                    br.fill();
                    cl[k_clen_order[k]] = (uint8_t)br.take(3);
                }
                uint16_t cc[8], csym[19];
                for (int l = 0; l < 8; l++)
                    cc[l] = 0;
                for (int k = 0; k < 19; k++)
                    cc[cl[k]]++;
                {
                    int left = 1;
                    for (int l = 1; l <= 7; l++) {
                        left <<= 1;
                        left -= cc[l];
                        if (left < 0)
                            status = PNG_ECODES;
                    }
                    uint16_t offs[8];
                    offs[1] = 0;
                    for (int l = 1; l < 7; l++)
                        offs[l + 1] = (uint16_t)(offs[l] + cc[l]);
                    for (int k = 0; k < 19; k++)
                        if (cl[k])
                            csym[offs[cl[k]]++] = (uint16_t)k;
                }
                int at = 0;
                while (status == PNG_OK && at < n_lit + n_dist) {
                    br.fill();
                    int code = 0, first = 0, index = 0, sym = -1;
                    for (int len = 1; len <= 7; len++) {
                        code |= (int)br.take(1);
                        const int c = cc[len];
                        if (code - c < first) {
                            sym = csym[index + (code - first)];
                            break;
                        }
                        index += c;
                        first += c;
                        first <<= 1;
                        code <<= 1;
                    }
                    br.fill();
                    if (sym < 0) {
                        status = PNG_ECODES;
                    } else if (sym < 16) {
                        L.lens[at++] = (uint8_t)sym;
                    } else {
                        int prev = 0, rep;
                        if (sym == 16) {
                            if (at == 0)
                                status = PNG_ECODES;
                            else
                                prev = L.lens[at - 1];
                            rep = 3 + (int)br.take(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)br.take(3);
                        } else {
                            rep = 11 + (int)br.take(7);
                        }
                        if (at + rep > n_lit + n_dist)
                            status = PNG_ECODES;
                        else
                            for (int k = 0; k < rep; k++)
                                L.lens[at++] = (uint8_t)prev;
                    }
                }
                if (status == PNG_OK && L.lens[256] == 0)
                    status = PNG_ECODES;   // no end-of-block code
                setup = 2;
            } else {
                status = PNG_EBLOCK;
            }
            continue;
        }

        if (phase == 2) {
            // ---- stored bytes (the reader is at a byte boundary)
            while (stored_left > 0 && br.next < limit && status == PNG_OK) {
                br.fill();
                if (op >= want) {
                    status = PNG_ELENGTH;
                    break;
                }
                out[op++] = (uint8_t)br.take(8);
                stored_left--;
            }
            if (stored_left == 0)
                phase = last ? 3 : 0;
            continue;
        }

        // ---- symbols of a coded block
        while (br.next < limit) {
            const int sym = decode_symbol(br, L.lit_fast, LIT_BITS, L.lit_count, L.lit_sym);
            if (sym < 0) {
                status = PNG_ESYMBOL;
                break;
            }
            if (sym < 256) {
                if (op >= want) {
                    status = PNG_ELENGTH;
                    break;
                }
                out[op++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) {
                phase = last ? 3 : 0;
                break;
            }
            if (sym > 285) {
                status = PNG_ESYMBOL;
                break;
            }
            br.fill();
            const int len = (int)k_len_base[sym - 257] + (int)br.take(k_len_extra[sym - 257]);
            const int ds = decode_symbol(br, L.dist_fast, DIST_BITS, L.dist_count, L.dist_sym);
            if (ds < 0 || ds > 29) {
                status = PNG_ESYMBOL;
                break;
            }
            br.fill();
            const int dist = (int)k_dist_base[ds] + (int)br.take(k_dist_extra[ds]);
            if ((int64_t)dist > op) {
                status = PNG_EDISTANCE;
                break;
            }
            if (op + len > want) {
                status = PNG_ELENGTH;
                break;
            }
            if (len >= COPY_TOGETHER) {
                copy_len = len;
                copy_dist = dist;
                break;
            }
            for (int k = 0; k < len; k++) {
                out[op] = out[op - dist];
                op++;
            }
        }
    }

    if (lane == 0) {
        if (status == PNG_OK) {
            // the Adler-32 of the output, big-endian, from the next byte boundary on (png_unfilter_kernel, which reads
            // every byte anyway, sums them and compares)
            br.drop(br.cnt & 7);
            uint32_t stored = 0;
            for (int k = 0; k < 4; k++) {
                br.fill();
                stored = (stored << 8) | br.take(8);
            }
            const int64_t consumed = (int64_t)br.next * 4 - (br.cnt >> 3);
            if (op != want)
                status = PNG_ELENGTH;
            else if (consumed > (int64_t)zlen)
                status = PNG_EINPUT;
            A.adler[f] = stored;
        }
        A.status[f] = status;
    }
}

__device__ __forceinline__ int png_paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// A wave per file: the rows one after the other (a row needs the row above), a row's bytes in LDS.  Filter types
// 0 (none) and 2 (up) are the lanes' work side by side; 1 (sub), 3 (average) and 4 (Paeth) run along the row, a lane
// per channel.  Then PIL's convert("L") of the row: (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16, alpha ignored.
__global__ __launch_bounds__(FX_WAVE) void png_unfilter_kernel(PngArgs A)
{
    __shared__ uint8_t rows[2][PNG_MAX_ROW];
    const int lane = fx_lane();
    const int64_t f = blockIdx.x;
    if (A.status[f] != PNG_OK)
        return;
    const int ch = A.channels[f];
    if (ch < 1 || ch > A.max_ch)   // (the inflate kernel has given such a file a header status; the rows in LDS hold max_ch channels)
        return;
    const int stride = A.W * ch;
    const uint8_t *raw = A.raw + A.raw_offset[f];
    uint8_t *out = A.out + f * (int64_t)A.H * A.W;
    int bad = 0;
    // Adler-32 of the inflated bytes b_0 .. b_{L-1}: low half 1 + sum b_i, high half L + sum (L - i) b_i, both
    // mod 65521 (each lane sums its bytes in 64 bits: at most 255 * L^2 / 64 < 2^63)
    const u64 L = (u64)A.H * (u64)(stride + 1);
    u64 s1 = 0, s2 = 0;
    for (int y = 0; y < A.H; y++) {
        uint8_t *cur = rows[y & 1];
        const uint8_t *up = rows[(y & 1) ^ 1];
        const uint8_t *src = raw + (int64_t)y * (stride + 1);
        const int ft = src[0];
        if (lane == 0) {
            s1 += (u64)ft;
            s2 += (L - (u64)y * (u64)(stride + 1)) * (u64)ft;
        }
        for (int i = lane; i < stride; i += FX_WAVE) {
            int v = src[1 + i];
            s1 += (u64)v;
            s2 += (L - ((u64)y * (u64)(stride + 1) + 1u + (u64)i)) * (u64)v;
            if (ft == 2 && y > 0)
                v += up[i];
            cur[i] = (uint8_t)v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ft == 1 || ft == 3 || ft == 4) {
            if (lane < ch) {
                int a = 0, c = 0;   // left, upper left
                for (int i = lane; i < stride; i += ch) {
                    const int b = y > 0 ? up[i] : 0;
                    const int x = cur[i];
                    const int pred = ft == 1 ? a : ft == 3 ? ((a + b) >> 1) : png_paeth(a, b, c);
                    a = (x + pred) & 255;
                    c = b;
                    cur[i] = (uint8_t)a;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else if (ft > 4) {
            bad = 1;
        }
        for (int x = lane; x < A.W; x += FX_WAVE) {
            const uint8_t *p = cur + x * ch;
            out[(int64_t)y * A.W + x] = ch <= 2 ? p[0] : (uint8_t)((p[0] * 19595u + p[1] * 38470u + p[2] * 7471u + 0x8000u) >> 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    s1 = fx_wave_sum_u64(s1);
    s2 = fx_wave_sum_u64(s2 % 65521ull);
    const uint32_t a1 = (uint32_t)((1ull + s1) % 65521ull), a2 = (uint32_t)((L + s2) % 65521ull);
    if (lane == 0) {
        if (bad)
            A.status[f] = PNG_EFILTER;
        else if (A.adler[f] != ((a2 << 16) | a1))
            A.status[f] = PNG_EADLER;
    }
}

// raw_offset[i] = sum over the files before i of H * (W * channels + 1): one block, files in chunks of 1024
__global__ __launch_bounds__(1024) void png_offsets_kernel(PngArgs A, int64_t *raw_offset, int64_t cap)
{
    __shared__ long long wave_tot[16];
    __shared__ long long carry;
    const int t = threadIdx.x, lane = fx_lane(), wave = fx_wave();
    if (t == 0)
        carry = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < A.n; i0 += 1024) {
        const int64_t i = i0 + t;
        const int ch = i < A.n ? A.channels[i] : 0;
        const long long mine = ch >= 1 && ch <= A.max_ch ? (long long)A.H * ((long long)A.W * ch + 1) : 0;
        long long inc = mine;
#pragma unroll
        for (int o = 1; o < FX_WAVE; o <<= 1) {
            const long long u = __shfl_up(inc, o, FX_WAVE);
            if (lane >= o)
                inc += u;
        }
        if (lane == FX_WAVE - 1)
            wave_tot[wave] = inc;
        __syncthreads();
        long long before = carry;
        for (int w = 0; w < wave; w++)
            before += wave_tot[w];
        if (i < A.n) {
            long long lo = before + inc - mine, hi = before + inc;
            // (a channel count that is not 1 .. max_ch, or a total beyond the scratch, gives the file an empty
            // stretch: its stream then ends with a length status)
            hi = hi > cap ? (lo > cap ? cap : lo) : hi;
            lo = lo > cap ? cap : lo;
            raw_offset[i] = lo;
            if (i == A.n - 1)
                raw_offset[A.n] = hi;
        }
        __syncthreads();
        if (t == 1023)
            carry = before + inc;
        __syncthreads();
    }
}

}  // namespace

extern "C" int fx_png_inflate_gray_u8(const uint8_t *d_staged, const int64_t *d_offset, const int32_t *d_length,
                                      const int32_t *d_channels, int64_t n, int H, int W, uint8_t *d_out, int32_t *d_status,
                                      void *stream)
{
    FX_REQUIRE(n >= 0 && n < (1ll << 31), "fx_png_inflate_gray_u8: n = %lld", (long long)n);
    if (n == 0)
        return FX_OK;
    FX_REQUIRE(H >= 1 && W >= 1 && W <= PNG_MAX_ROW, "fx_png_inflate_gray_u8: %d x %d", H, W);
    FX_REQUIRE(d_staged && d_offset && d_length && d_channels && d_out && d_status, "fx_png_inflate_gray_u8: null pointer");
    hipStream_t s = (hipStream_t)stream;
    FxCallGuard guard(s);
    // unfiltered rows of every file: at most four channels
    int max_ch = PNG_MAX_ROW / W;
    max_ch = max_ch > 4 ? 4 : max_ch;
    FX_REQUIRE(max_ch >= 1, "fx_png_inflate_gray_u8: rows of %d pixels are too long", W);
    const int64_t cap = n * (int64_t)H * ((int64_t)W * max_ch + 1);
    const size_t off_bytes = (((size_t)n + 1) * sizeof(int64_t) + 255) & ~(size_t)255;
    const size_t adler_bytes = ((size_t)n * sizeof(uint32_t) + 255) & ~(size_t)255;
    unsigned char *base = (unsigned char *)guard.scratch(off_bytes + adler_bytes + (size_t)cap, 3);
    if (!base)
        return FX_ENOMEM;
    PngArgs A;
    A.staged = d_staged;
    A.offset = d_offset;
    A.length = d_length;
    A.channels = d_channels;
    A.raw_offset = (const int64_t *)base;
    A.n = n;
    A.H = H;
    A.W = W;
    A.max_ch = max_ch;
    A.adler = (uint32_t *)(base + off_bytes);
    A.raw = base + off_bytes + adler_bytes;
    A.out = d_out;
    A.status = d_status;
    hipLaunchKernelGGL(png_offsets_kernel, dim3(1), dim3(1024), 0, s, A, (int64_t *)base, cap);
    FxProfileScope prof(s);
    hipLaunchKernelGGL(png_inflate_kernel, dim3((unsigned)n), dim3(FX_WAVE), 0, s, A);
    hipLaunchKernelGGL(png_unfilter_kernel, dim3((unsigned)n), dim3(FX_WAVE), 0, s, A);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}
