// rans.hip -- host-side entropy coding of the factorized-prior symbols: a 64-bit range-variant ANS
// coder with 32-bit renormalisation, 16-bit probabilities and a 4-bit bypass mode for out-of-table symbols, and the
// PMF -> quantised CDF routine.  Bit-exact replacement of the reference's pybind11 extensions
//   compressai/cpp_exts/rans/rans_interface.cpp:108-283 (BufferedRansEncoder / RansEncoder / RansDecoder, built there on
//   third_party/ryg_rans/rans64.h, F. Giesen's public-domain rANS) and
//   compressai/cpp_exts/ops/ops.cpp:41-106 (pmf_to_quantized_cdf),
// behind a plain C ABI (flat int32 tables instead of vector<vector<int>>, caller-owned byte buffers).  The symbol streams
// come from the device (masic_symbols_fwd); coding itself is sequential by construction and stays on the host, as in the
// reference.  Wire format: the encoder's words in the order the decoder reads them (two state words, then the
// renormalisation words), little endian.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace {

constexpr int kPrecision = 16;                 // probability resolution of the tables
constexpr int kBypassBits = 4;                 // width of one bypass digit
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;
constexpr uint64_t kLow = 1ull << 31;          // lower bound of the normalised state interval

struct Sym {
    uint16_t start, range;
    bool bypass;
};

// x = C(s, x) for a table symbol [start, start + freq) out of 2^16
inline void put(uint64_t& x, uint32_t*& p, uint32_t start, uint32_t freq) {
    const uint64_t x_max = ((kLow >> kPrecision) << 32) * freq;
    if (x >= x_max) {
        *--p = (uint32_t)x;
        x >>= 32;
    }
    x = ((x / freq) << kPrecision) + (x % freq) + start;
}

// raw digit of `nbits` bits (a uniform symbol of frequency 2^(16-nbits))
inline void put_bits(uint64_t& x, uint32_t*& p, uint32_t val, uint32_t nbits) {
    const uint64_t x_max = ((kLow >> 16) << 32) * (uint64_t)(1u << (16 - nbits));
    if (x >= x_max) {
        *--p = (uint32_t)x;
        x >>= 32;
    }
    x = (x << nbits) | val;
}

inline uint32_t get_bits(uint64_t& x, const uint32_t*& p, const uint32_t* end, uint32_t nbits, bool& ok) {
    const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < kLow) {
        if (p >= end) { ok = false; return val; }
        x = (x << 32) | *p++;
    }
    return val;
}

int check_tables(const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes, int ncdfs) {
    for (int i = 0; i < ncdfs; ++i) {
        const int n = cdf_sizes[i];
        MASIC_REQUIRE(n >= 2 && n <= cdf_stride, MASIC_ERR_SHAPE, "rans: cdf %d has size %d (stride %d)", i, n, cdf_stride);
        const int32_t* c = cdfs + (size_t)i * cdf_stride;
        MASIC_REQUIRE(c[0] == 0 && c[n - 1] == (1 << kPrecision), MASIC_ERR_ARG, "rans: cdf %d does not span [0, 2^16]", i);
        for (int j = 0; j + 1 < n; ++j) MASIC_REQUIRE(c[j + 1] > c[j], MASIC_ERR_ARG, "rans: cdf %d is not strictly increasing at %d", i, j);
    }
    return MASIC_OK;
}

}  // namespace

extern "C" int masic_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf) {
    MASIC_REQUIRE(pmf && cdf && n > 0 && precision > 0 && precision <= 16, MASIC_ERR_ARG, "pmf_to_quantized_cdf: bad argument");
    for (int i = 0; i < n; ++i)
        MASIC_REQUIRE(pmf[i] >= 0.0f && isfinite(pmf[i]), MASIC_ERR_ARG, "pmf_to_quantized_cdf: non-finite or negative element %g", (double)pmf[i]);
    // frequencies: round(p * 2^precision) (float arithmetic, round half away from zero), rescaled to sum to at most 2^precision
    cdf[0] = 0;
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) {
        cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));
        total += cdf[i + 1];
    }
    MASIC_REQUIRE(total != 0, MASIC_ERR_ARG, "pmf_to_quantized_cdf: all probabilities are zero");
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)(((uint64_t)(1u << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    // every symbol needs a non-zero frequency: take one count from the rarest symbol that can spare it
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        MASIC_REQUIRE(best >= 0, MASIC_ERR_ARG, "pmf_to_quantized_cdf: more symbols than counts");
        if (best < i) {
            for (int j = best + 1; j <= i; ++j) cdf[j]--;
        } else {
            for (int j = i + 1; j <= best; ++j) cdf[j]++;
        }
    }
    return MASIC_OK;
}

extern "C" size_t masic_rans_encode_bound(int nsymbols) {
    // a symbol costs at most one 32-bit word per coder event; an escaped symbol adds at most 10 events (count + 8 digits)
    return ((size_t)nsymbols * 11 + 2) * sizeof(uint32_t);
}

extern "C" int masic_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int n, const int32_t* cdfs, int cdf_stride,
                                              const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, uint8_t* out, size_t out_cap,
                                              size_t* out_len) {
    MASIC_REQUIRE(symbols && indexes && cdfs && cdf_sizes && offsets && out && out_len && n >= 0, MASIC_ERR_ARG, "rans_encode: null pointer");
    int rc = check_tables(cdfs, cdf_stride, cdf_sizes, ncdfs);
    if (rc != MASIC_OK) return rc;
    // forward pass: symbol -> coder events (table symbol, then for the sentinel the escaped value as bypass digits)
    std::vector<Sym> ev;
    ev.reserve((size_t)n + 16);
    for (int i = 0; i < n; ++i) {
        const int idx = indexes[i];
        MASIC_REQUIRE(idx >= 0 && idx < ncdfs, MASIC_ERR_ARG, "rans_encode: index %d out of range at %d", idx, i);
        const int32_t* cdf = cdfs + (size_t)idx * cdf_stride;
        const int32_t max_value = cdf_sizes[idx] - 2;
        int32_t value = symbols[i] - offsets[idx];
        uint32_t raw = 0;
        if (value < 0) {
            raw = (uint32_t)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw = (uint32_t)(2 * (value - max_value));
            value = max_value;
        }
        ev.push_back({(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), false});
        if (value == max_value) {
            int32_t nb = 0;                                  // 4-bit digits of the escaped value (at most 8: a 32-bit shift is not a shift)
            while (nb < 8 && (raw >> (nb * kBypassBits)) != 0) ++nb;
            int32_t v = nb;
            while (v >= (int32_t)kBypassMax) {
                ev.push_back({(uint16_t)kBypassMax, (uint16_t)(kBypassMax + 1), true});
                v -= kBypassMax;
            }
            ev.push_back({(uint16_t)v, (uint16_t)(v + 1), true});
            for (int32_t j = 0; j < nb; ++j) {
                const uint32_t dgt = (raw >> (j * kBypassBits)) & kBypassMax;
                ev.push_back({(uint16_t)dgt, (uint16_t)(dgt + 1), true});
            }
        }
    }
    // backward pass: the coder is a stack -- events are pushed last to first, words are written from the end of the buffer
    std::vector<uint32_t> buf(ev.size() + 2);
    uint32_t* p = buf.data() + buf.size();
    uint64_t x = kLow;
    for (size_t k = ev.size(); k-- > 0;) {
        const Sym& s = ev[k];
        if (s.bypass) put_bits(x, p, s.start, kBypassBits);
        else put(x, p, s.start, s.range);
    }
    p -= 2;
    p[0] = (uint32_t)x;
    p[1] = (uint32_t)(x >> 32);
    const size_t nbytes = (size_t)(buf.data() + buf.size() - p) * sizeof(uint32_t);
    MASIC_REQUIRE(nbytes <= out_cap, MASIC_ERR_SHAPE, "rans_encode: output buffer of %zu bytes, %zu needed", out_cap, nbytes);
    memcpy(out, p, nbytes);
    *out_len = nbytes;
    return MASIC_OK;
}

namespace {
// symbols i = 0 .. n-1 with tables picked by indexes[i], continuing from the coder state (x, p): shared by the one-shot and the
// streaming decoder (reference rans_interface.cpp:214-283 and :286-353 -- the same loop over a stream kept between calls)
int decode_indexed(uint64_t& x, const uint32_t*& p, const uint32_t* end, const int32_t* indexes, int n, const int32_t* cdfs, int cdf_stride,
                   const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, int32_t* symbols) {
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        const int idx = indexes[i];
        MASIC_REQUIRE(idx >= 0 && idx < ncdfs, MASIC_ERR_ARG, "rans_decode: index %d out of range at %d", idx, i);
        const int32_t* cdf = cdfs + (size_t)idx * cdf_stride;
        const int32_t size = cdf_sizes[idx], max_value = size - 2;
        const uint32_t cum = (uint32_t)(x & ((1u << kPrecision) - 1));
        // first table entry above cum (tables are short: the reference scans linearly too; here a binary search)
        int lo = 0, hi = size - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((uint32_t)cdf[mid] > cum) hi = mid;
            else lo = mid + 1;
        }
        const int s = lo - 1;
        MASIC_REQUIRE(s >= 0 && s <= max_value, MASIC_ERR_ARG, "rans_decode: corrupt stream at symbol %d", i);
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        x = (uint64_t)freq * (x >> kPrecision) + (x & ((1u << kPrecision) - 1)) - start;
        if (x < kLow) {
            MASIC_REQUIRE(p < end, MASIC_ERR_ARG, "rans_decode: stream ends at symbol %d", i);
            x = (x << 32) | *p++;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t v = (int32_t)get_bits(x, p, end, kBypassBits, ok);
            int32_t nb = v;
            while (ok && v == (int32_t)kBypassMax) {
                v = (int32_t)get_bits(x, p, end, kBypassBits, ok);
                nb += v;
            }
            uint32_t raw = 0;
            for (int32_t j = 0; ok && j < nb; ++j) raw |= get_bits(x, p, end, kBypassBits, ok) << (j * kBypassBits);
            MASIC_REQUIRE(ok, MASIC_ERR_ARG, "rans_decode: stream ends inside an escaped symbol at %d", i);
            value = (int32_t)(raw >> 1);
            if (raw & 1) value = -value - 1;
            else value += max_value;
        }
        symbols[i] = value + offsets[idx];
    }
    return MASIC_OK;
}
}  // namespace

extern "C" int masic_rans_decode_with_indexes(const uint8_t* in, size_t in_len, const int32_t* indexes, int n, const int32_t* cdfs,
                                              int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, int32_t* symbols) {
    MASIC_REQUIRE(in && indexes && cdfs && cdf_sizes && offsets && symbols && n >= 0, MASIC_ERR_ARG, "rans_decode: null pointer");
    MASIC_REQUIRE(in_len >= 8 && in_len % 4 == 0, MASIC_ERR_SHAPE, "rans_decode: stream of %zu bytes", in_len);
    int rc = check_tables(cdfs, cdf_stride, cdf_sizes, ncdfs);
    if (rc != MASIC_OK) return rc;
    std::vector<uint32_t> words(in_len / 4);
    memcpy(words.data(), in, in_len);
    const uint32_t* p = words.data();
    const uint32_t* end = p + words.size();
    uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    p += 2;
    return decode_indexed(x, p, end, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, ncdfs, symbols);
}

// ---------------------------------------------------------------------------- adaptive tables
// The y streams of HSIC.compress (reference MASIC.py:855-1158) carry one table PER SYMBOL -- the Gaussian-mixture PMF of
// that latent element given its context -- so the coder takes (start, freq) pairs instead of table indexes.  The reference
// drives the third-party `range_coder` package here (not in the image: its byte stream is parity-unpinned); this container
// uses the rANS coder above with the same 16-bit resolution.  Decoding is incremental: the tables of a wavefront of
// symbols only exist once the previous wavefront is decoded.
extern "C" int masic_rans_encode_freqs(const int32_t* start_freq, size_t n, uint8_t* out, size_t out_cap, size_t* out_len) {
    MASIC_REQUIRE(start_freq && out && out_len, MASIC_ERR_ARG, "rans_encode_freqs: null pointer");
    std::vector<uint32_t> buf(n + 2);
    uint32_t* p = buf.data() + buf.size();
    uint64_t x = kLow;
    for (size_t k = n; k-- > 0;) {
        const int32_t st = start_freq[2 * k], fr = start_freq[2 * k + 1];
        MASIC_REQUIRE(st >= 0 && fr >= 1 && st + fr <= (1 << kPrecision), MASIC_ERR_ARG, "rans_encode_freqs: symbol %zu has interval [%d, %d + %d)", k, st, st, fr);
        put(x, p, (uint32_t)st, (uint32_t)fr);
    }
    p -= 2;
    p[0] = (uint32_t)x;
    p[1] = (uint32_t)(x >> 32);
    const size_t nbytes = (size_t)(buf.data() + buf.size() - p) * sizeof(uint32_t);
    MASIC_REQUIRE(nbytes <= out_cap, MASIC_ERR_SHAPE, "rans_encode_freqs: output buffer of %zu bytes, %zu needed", out_cap, nbytes);
    memcpy(out, p, nbytes);
    *out_len = nbytes;
    return MASIC_OK;
}

// One rANS stream PER CHANNEL (the device decoder of codec.hip runs one wavefront per stream): start_freq is [npix * nch][2], row i * nch + c =
// symbol of pixel i (coding order) and channel c.  out: the nch streams back to back, lengths[c] bytes each (multiples of 4, >= 8).
extern "C" int masic_rans_encode_channels(const int32_t* start_freq, int npix, int nch, uint8_t* out, size_t out_cap, uint32_t* lengths, size_t* out_len) {
    MASIC_REQUIRE(start_freq && out && lengths && out_len && npix >= 0 && nch >= 1, MASIC_ERR_ARG, "rans_encode_channels: bad argument");
    std::vector<uint32_t> buf((size_t)npix + 2);
    size_t total = 0;
    for (int c = 0; c < nch; ++c) {
        uint32_t* p = buf.data() + buf.size();
        uint64_t x = kLow;
        for (int k = npix; k-- > 0;) {
            const int32_t st = start_freq[2 * ((size_t)k * nch + c)], fr = start_freq[2 * ((size_t)k * nch + c) + 1];
            MASIC_REQUIRE(st >= 0 && fr >= 1 && st + fr <= (1 << kPrecision), MASIC_ERR_ARG, "rans_encode_channels: pixel %d channel %d has interval [%d, %d + %d)", k, c, st, st, fr);
            put(x, p, (uint32_t)st, (uint32_t)fr);
        }
        p -= 2;
        p[0] = (uint32_t)x;
        p[1] = (uint32_t)(x >> 32);
        const size_t nbytes = (size_t)(buf.data() + buf.size() - p) * sizeof(uint32_t);
        MASIC_REQUIRE(total + nbytes <= out_cap, MASIC_ERR_SHAPE, "rans_encode_channels: output buffer of %zu bytes too small", out_cap);
        memcpy(out + total, p, nbytes);
        lengths[c] = (uint32_t)nbytes;
        total += nbytes;
    }
    *out_len = total;
    return MASIC_OK;
}

namespace {
struct AdaptiveDecoder {
    std::vector<uint32_t> words;
    size_t pos;
    uint64_t x;
};
}  // namespace

extern "C" int masic_rans_decoder_open(const uint8_t* in, size_t in_len, void** handle) {
    MASIC_REQUIRE(in && handle, MASIC_ERR_ARG, "rans_decoder_open: null pointer");
    MASIC_REQUIRE(in_len >= 8 && in_len % 4 == 0, MASIC_ERR_SHAPE, "rans_decoder_open: stream of %zu bytes", in_len);
    AdaptiveDecoder* d = new AdaptiveDecoder();
    d->words.resize(in_len / 4);
    memcpy(d->words.data(), in, in_len);
    d->x = (uint64_t)d->words[0] | ((uint64_t)d->words[1] << 32);
    d->pos = 2;
    *handle = d;
    return MASIC_OK;
}

// rows of L interval starts (u16, starts[0] = 0, strictly increasing, the last interval ends at 2^16); one symbol per row
extern "C" int masic_rans_decoder_decode_rows(void* handle, const uint16_t* starts, int nrows, int L, int32_t* symbols) {
    MASIC_REQUIRE(handle && starts && symbols && nrows >= 0 && L >= 1, MASIC_ERR_ARG, "rans_decoder_decode_rows: bad argument");
    AdaptiveDecoder* d = (AdaptiveDecoder*)handle;
    for (int r = 0; r < nrows; ++r) {
        const uint16_t* c = starts + (size_t)r * L;
        const uint32_t cum = (uint32_t)(d->x & ((1u << kPrecision) - 1));
        int lo = 0, hi = L;                                  // first start above cum
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((uint32_t)c[mid] > cum) hi = mid;
            else lo = mid + 1;
        }
        const int s = lo - 1;
        MASIC_REQUIRE(s >= 0, MASIC_ERR_ARG, "rans_decoder_decode_rows: table of row %d does not start at 0", r);
        const uint32_t st = c[s], en = s + 1 < L ? (uint32_t)c[s + 1] : (1u << kPrecision);
        MASIC_REQUIRE(en > st, MASIC_ERR_ARG, "rans_decoder_decode_rows: empty interval in row %d", r);
        d->x = (uint64_t)(en - st) * (d->x >> kPrecision) + cum - st;
        if (d->x < kLow) {
            MASIC_REQUIRE(d->pos < d->words.size(), MASIC_ERR_ARG, "rans_decoder_decode_rows: stream ends at row %d", r);
            d->x = (d->x << 32) | d->words[d->pos++];
        }
        symbols[r] = s;
    }
    return MASIC_OK;
}

// streaming form of masic_rans_decode_with_indexes on a decoder opened with masic_rans_decoder_open: decodes the next n symbols and
// keeps the coder state for the next call (reference RansDecoder::set_stream / decode_stream, rans_interface.cpp:286-353)
extern "C" int masic_rans_decoder_decode_indexes(void* handle, const int32_t* indexes, int n, const int32_t* cdfs, int cdf_stride,
                                                 const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, int32_t* symbols) {
    MASIC_REQUIRE(handle && indexes && cdfs && cdf_sizes && offsets && symbols && n >= 0, MASIC_ERR_ARG, "rans_decoder_decode_indexes: null pointer");
    int rc = check_tables(cdfs, cdf_stride, cdf_sizes, ncdfs);
    if (rc != MASIC_OK) return rc;
    AdaptiveDecoder* d = (AdaptiveDecoder*)handle;
    const uint32_t* p = d->words.data() + d->pos;
    const uint32_t* end = d->words.data() + d->words.size();
    rc = decode_indexed(d->x, p, end, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, ncdfs, symbols);
    d->pos = (size_t)(p - d->words.data());
    return rc;
}

extern "C" void masic_rans_decoder_close(void* handle) { delete (AdaptiveDecoder*)handle; }
