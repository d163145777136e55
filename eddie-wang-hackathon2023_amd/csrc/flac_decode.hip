// FLAC stream decoder (host code): the reference shells out to ffmpeg for every audio file
// (W/whisper_utils.py:17-54); the MI355X boxes have no ffmpeg, and LibriSpeech ships FLAC, so the
// library carries its own decoder.  Covers the whole subset format and more: CONSTANT / VERBATIM / FIXED /
// LPC subframes, Rice + Rice2 residuals with escape partitions, wasted bits, all four channel
// assignments, 4..32 bit samples, fixed and variable block size.  Every frame's CRC-8 (header) and CRC-16
// (whole frame) are checked; the STREAMINFO MD5 is returned so that the caller can verify the PCM.
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"
#include "whisper_mi355.h"

namespace wm {
namespace {

struct BitReader {
    const uint8_t* p; size_t n; size_t pos = 0;      // byte cursor
    uint64_t acc = 0; int nacc = 0;                  // bit accumulator (msb first)
    bool overrun = false;
    BitReader(const uint8_t* data, size_t bytes) : p(data), n(bytes) {}
    inline void refill() {
        while (nacc <= 56) {
            uint64_t b = 0;
            if (pos < n) b = p[pos]; else overrun = pos > n + 8 ? true : overrun;
            ++pos;
            acc |= b << (56 - nacc);
            nacc += 8;
        }
    }
    inline uint32_t bits(int k) {                    // k in [0, 32]
        if (k == 0) return 0;
        if (nacc < k) refill();
        const uint32_t v = (uint32_t)(acc >> (64 - k));
        acc <<= k; nacc -= k;
        return v;
    }
    inline int32_t sbits(int k) {
        if (k == 0) return 0;
        const uint32_t v = bits(k);
        return (int32_t)(v << (32 - k)) >> (32 - k);
    }
    inline uint32_t unary() {                        // number of 0 bits before the next 1 bit
        uint32_t q = 0;
        for (;;) {
            if (nacc == 0) refill();
            if (acc == 0) {
                q += nacc; nacc = 0; acc = 0;
                if (pos > n + 8) { overrun = true; return q; }
                continue;
            }
            const int z = __builtin_clzll(acc);
            if (z < nacc) { q += z; acc <<= (z + 1); nacc -= z + 1; return q; }
            q += nacc; nacc = 0; acc = 0;
        }
    }
    inline void align() { const int r = nacc & 7; acc <<= r; nacc -= r; }
    inline size_t byte_pos() const { return pos - (size_t)(nacc >> 3); }   // valid when byte aligned
    inline bool past_end() const { return byte_pos() > n; }
};

uint8_t crc8(const uint8_t* d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= d[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
    }
    return c;
}

uint16_t crc16(const uint8_t* d, size_t n) {
    static uint16_t table[256];
    static bool ready = false;
    if (!ready) {
        for (int i = 0; i < 256; ++i) {
            uint16_t c = (uint16_t)(i << 8);
            for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
            table[i] = c;
        }
        ready = true;
    }
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ table[(c >> 8) ^ d[i]]);
    return c;
}

struct StreamInfo {
    int min_block = 0, max_block = 0, sample_rate = 0, channels = 0, bps = 0;
    int64_t total = 0;
    uint8_t md5[16] = {0};
    size_t first_frame = 0;                          // byte offset of the first audio frame
};

int parse_header(const uint8_t* d, size_t n, StreamInfo& si) {
    WM_REQUIRE(n >= 42 && memcmp(d, "fLaC", 4) == 0, "flac: not a FLAC stream (no fLaC marker)");
    size_t off = 4;
    bool have_info = false;
    for (;;) {
        WM_REQUIRE(off + 4 <= n, "flac: truncated metadata");
        const bool last = d[off] & 0x80;
        const int type = d[off] & 0x7f;
        const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
        off += 4;
        WM_REQUIRE(off + len <= n, "flac: truncated metadata block (type %d, %zu bytes)", type, len);
        if (type == 0) {
            WM_REQUIRE(len >= 34, "flac: STREAMINFO too short");
            const uint8_t* s = d + off;
            si.min_block = (s[0] << 8) | s[1];
            si.max_block = (s[2] << 8) | s[3];
            si.sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si.channels = ((s[12] >> 1) & 7) + 1;
            si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si.total = ((int64_t)(s[13] & 15) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
            memcpy(si.md5, s + 18, 16);
            have_info = true;
        }
        off += len;
        if (last) break;
    }
    WM_REQUIRE(have_info, "flac: no STREAMINFO block");
    si.first_frame = off;
    return 0;
}

int decode_residual(BitReader& br, int32_t* out, int block, int order) {
    const int method = br.bits(2);
    WM_REQUIRE(method < 2, "flac: reserved residual coding method %d", method);
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = br.bits(4);
    const int parts = 1 << porder;
    WM_REQUIRE((block >> porder) << porder == block || porder == 0, "flac: block %d not divisible by 2^%d", block, porder);
    WM_REQUIRE((block >> porder) >= order || porder == 0, "flac: partition smaller than predictor order");
    int i = order;
    for (int part = 0; part < parts; ++part) {
        int count = porder ? (block >> porder) : block;
        if (part == 0) count -= order;
        const int k = br.bits(pbits);
        if (k == esc) {
            const int raw = br.bits(5);
            for (int j = 0; j < count; ++j) out[i++] = br.sbits(raw);
        } else {
            for (int j = 0; j < count; ++j) {
                const uint32_t q = br.unary();
                const uint32_t u = (q << k) | br.bits(k);
                out[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        WM_REQUIRE(!br.overrun, "flac: ran past the end of the stream inside a residual");
    }
    return 0;
}

int decode_subframe(BitReader& br, int32_t* out, int block, int bps) {
    WM_REQUIRE(br.bits(1) == 0, "flac: subframe padding bit set");
    const int type = br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    WM_REQUIRE(wasted < bps, "flac: %d wasted bits in a %d-bit subframe", wasted, bps);
    bps -= wasted;
    auto wide = [&](int b) -> int32_t {              // side channels of 32-bit streams need 33 bits: not supported
        return br.sbits(b);
    };
    WM_REQUIRE(bps <= 32, "flac: %d-bit subframe not supported", bps);
    if (type == 0) {
        const int32_t v = wide(bps);
        for (int i = 0; i < block; ++i) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < block; ++i) out[i] = wide(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        WM_REQUIRE(order <= block, "flac: fixed order %d > block %d", order, block);
        for (int i = 0; i < order; ++i) out[i] = wide(bps);
        if (decode_residual(br, out, block, order)) return 1;
        switch (order) {
            case 1: for (int i = 1; i < block; ++i) out[i] += out[i - 1]; break;
            case 2: for (int i = 2; i < block; ++i) out[i] += 2 * out[i - 1] - out[i - 2]; break;
            case 3: for (int i = 3; i < block; ++i) out[i] += 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
            case 4: for (int i = 4; i < block; ++i) out[i] += 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
            default: break;
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        WM_REQUIRE(order <= block, "flac: LPC order %d > block %d", order, block);
        for (int i = 0; i < order; ++i) out[i] = wide(bps);
        const int precision = br.bits(4) + 1;
        WM_REQUIRE(precision != 16, "flac: reserved LPC precision");
        const int shift = br.sbits(5);
        WM_REQUIRE(shift >= 0, "flac: negative LPC shift");
        int32_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.sbits(precision);
        if (decode_residual(br, out, block, order)) return 1;
        for (int i = order; i < block; ++i) {
            int64_t pred = 0;
            for (int j = 0; j < order; ++j) pred += (int64_t)coef[j] * out[i - 1 - j];
            out[i] += (int32_t)(pred >> shift);
        }
    } else {
        WM_REQUIRE(false, "flac: reserved subframe type %d", type);
    }
    if (wasted)
        for (int i = 0; i < block; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
    return 0;
}

int decode_stream(const uint8_t* d, size_t n, const StreamInfo& si, int32_t* pcm, int64_t capacity, int64_t* n_out) {
    static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int kBits[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    std::vector<int32_t> ch[8];
    int64_t done = 0;
    size_t off = si.first_frame;
    while (off + 2 <= n && (si.total == 0 || done < si.total)) {
        WM_REQUIRE(d[off] == 0xff && (d[off + 1] & 0xfc) == 0xf8, "flac: lost frame sync at byte %zu", off);
        BitReader br(d + off, n - off);
        br.bits(14); br.bits(1);
        br.bits(1);                                          // blocking strategy: only changes what the coded number counts
        const int bcode = br.bits(4), rcode = br.bits(4), assign = br.bits(4), scode = br.bits(3);
        WM_REQUIRE(br.bits(1) == 0, "flac: reserved frame header bit set");
        {   // UTF-8 style coded frame / sample number (up to 36 bits): skip it
            int lead = br.bits(8), extra = 0;
            while (lead & 0x80) { lead <<= 1; ++extra; }
            WM_REQUIRE(extra != 1 && extra <= 7, "flac: bad coded number");
            for (int i = 1; i < extra; ++i) br.bits(8);
        }
        int block = kBlock[bcode];
        if (bcode == 6) block = br.bits(8) + 1;
        else if (bcode == 7) block = br.bits(16) + 1;
        WM_REQUIRE(block > 0, "flac: reserved block size code");
        if (rcode == 12) br.bits(8); else if (rcode == 13 || rcode == 14) br.bits(16);
        WM_REQUIRE(rcode != 15, "flac: invalid sample rate code");
        const size_t hdr_len = br.byte_pos();
        const int hdr_crc = br.bits(8);
        WM_REQUIRE(off + hdr_len < n && crc8(d + off, hdr_len) == hdr_crc, "flac: frame header CRC mismatch at byte %zu", off);
        WM_REQUIRE(scode != 3, "flac: reserved sample size code");
        const int bps = scode ? kBits[scode] : si.bps;
        const int nch = assign < 8 ? assign + 1 : 2;
        WM_REQUIRE(assign <= 10, "flac: reserved channel assignment %d", assign);
        WM_REQUIRE(nch == si.channels, "flac: frame has %d channels, stream has %d", nch, si.channels);
        for (int c = 0; c < nch; ++c) {
            if ((int)ch[c].size() < block) ch[c].resize(block);
            const bool side = (assign == 8 && c == 1) || (assign == 9 && c == 0) || (assign == 10 && c == 1);
            if (decode_subframe(br, ch[c].data(), block, bps + (side ? 1 : 0))) return 1;
        }
        br.align();
        const size_t body_len = br.byte_pos();
        const int frame_crc = br.bits(16);
        WM_REQUIRE(!br.overrun && off + body_len + 2 <= n, "flac: truncated frame at byte %zu", off);
        WM_REQUIRE(crc16(d + off, body_len) == frame_crc, "flac: frame CRC-16 mismatch at byte %zu", off);
        off += body_len + 2;

        if (assign == 8) {
            for (int i = 0; i < block; ++i) ch[1][i] = ch[0][i] - ch[1][i];
        } else if (assign == 9) {
            for (int i = 0; i < block; ++i) ch[0][i] += ch[1][i];
        } else if (assign == 10) {
            for (int i = 0; i < block; ++i) {
                const int32_t side = ch[1][i];
                const int32_t mid = (int32_t)(((uint32_t)ch[0][i] << 1) | (side & 1));
                ch[0][i] = (mid + side) >> 1;
                ch[1][i] = (mid - side) >> 1;
            }
        }
        int64_t take = block;
        if (si.total && done + take > si.total) take = si.total - done;
        WM_REQUIRE(done + take <= capacity, "flac: output buffer holds %lld samples per channel, stream has more", (long long)capacity);
        for (int c = 0; c < nch; ++c) {
            int32_t* dst = pcm + done * nch + c;
            const int32_t* src = ch[c].data();
            for (int64_t i = 0; i < take; ++i) dst[i * nch] = src[i];
        }
        done += take;
    }
    WM_REQUIRE(si.total == 0 || done == si.total, "flac: decoded %lld of %lld samples", (long long)done, (long long)si.total);
    *n_out = done;
    return 0;
}

}  // namespace
}  // namespace wm

extern "C" {

int wm_flac_info(const void* data, size_t bytes, wm_flac_streaminfo* info) {
    using namespace wm;
    WM_REQUIRE(data != nullptr && info != nullptr, "flac: null argument");
    StreamInfo si;
    if (parse_header((const uint8_t*)data, bytes, si)) return 1;
    info->sample_rate = si.sample_rate;
    info->channels = si.channels;
    info->bits_per_sample = si.bps;
    info->max_block_size = si.max_block;
    info->total_samples = si.total;
    memcpy(info->md5, si.md5, 16);
    return 0;
}

int wm_flac_decode(const void* data, size_t bytes, int32_t* pcm, int64_t capacity_samples, int64_t* n_decoded) {
    using namespace wm;
    WM_REQUIRE(data != nullptr && pcm != nullptr && n_decoded != nullptr, "flac: null argument");
    StreamInfo si;
    if (parse_header((const uint8_t*)data, bytes, si)) return 1;
    return decode_stream((const uint8_t*)data, bytes, si, pcm, capacity_samples, n_decoded);
}

}  // extern "C"
