// Host side of the input pipeline's file decoding (reference: dataset/multi_speaker_dataset.py:15-19
// `librosa.load(path, sr=16000)`): a RIFF/WAVE reader that returns what librosa returns BEFORE resampling - float32 mono, integer PCM
// scaled by 2^-(bits-1) (libsndfile's law: int16 / 32768, 24-bit / 8388608, int32 / 2147483648, uint8 (x - 128) / 128), IEEE float as
// stored, channels averaged in float32 (librosa.to_mono = np.mean over the channel axis).  The resampling to 16 kHz, the slicing, mixing
// and mask generation run on the device (preprocess.hip).  These two functions take HOST pointers: they are file I/O, not kernels.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "av_common.h"

namespace {

struct WavFmt {
    int tag = 0, channels = 0, rate = 0, bits = 0, block = 0;
    long long data_off = 0, data_bytes = 0;
};

unsigned rd_u32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
unsigned rd_u16(const unsigned char* p) { return p[0] | (p[1] << 8); }

// walks the RIFF chunks; returns 0 on success
int parse_wav(FILE* f, WavFmt& w, const char* path) {
    unsigned char h[12];
    if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) {
        av_set_error("av_wav: %s is not a RIFF/WAVE file", path);
        return AV_ERR_ARG;
    }
    bool have_fmt = false;
    for (;;) {
        unsigned char ch[8];
        if (fread(ch, 1, 8, f) != 8) break;
        const unsigned sz = rd_u32(ch + 4);
        if (!memcmp(ch, "fmt ", 4)) {
            unsigned char b[40] = {0};
            const unsigned n = sz < 40 ? sz : 40;
            if (sz < 16 || fread(b, 1, n, f) != n) { av_set_error("av_wav: %s: short fmt chunk", path); return AV_ERR_ARG; }
            w.tag = (int)rd_u16(b); w.channels = (int)rd_u16(b + 2); w.rate = (int)rd_u32(b + 4); w.block = (int)rd_u16(b + 12); w.bits = (int)rd_u16(b + 14);
            if (w.tag == 0xFFFE && sz >= 26) w.tag = (int)rd_u16(b + 24);          // WAVE_FORMAT_EXTENSIBLE: first two bytes of the sub-format GUID
            if (fseek(f, (long)(sz - n) + (sz & 1), SEEK_CUR)) break;
            have_fmt = true;
        } else if (!memcmp(ch, "data", 4)) {
            w.data_off = ftell(f);
            w.data_bytes = sz;
            if (!have_fmt) { av_set_error("av_wav: %s: data chunk before fmt", path); return AV_ERR_ARG; }
            // a streamed file may carry 0 / 0xFFFFFFFF here: use what is actually there
            fseek(f, 0, SEEK_END);
            const long long avail = ftell(f) - w.data_off;
            if (w.data_bytes == 0 || w.data_bytes == 0xFFFFFFFFu || w.data_bytes > avail) w.data_bytes = avail;
            return AV_OK;
        } else {
            if (fseek(f, (long)sz + (sz & 1), SEEK_CUR)) break;
        }
    }
    av_set_error("av_wav: %s: no data chunk", path);
    return AV_ERR_ARG;
}

int check_fmt(const WavFmt& w, const char* path) {
    const bool pcm = w.tag == 1 && (w.bits == 8 || w.bits == 16 || w.bits == 24 || w.bits == 32);
    const bool flt = w.tag == 3 && (w.bits == 32 || w.bits == 64);
    if (!(pcm || flt) || w.channels < 1 || w.channels > 64 || w.rate < 1 || w.block != w.channels * w.bits / 8) {
        av_set_error("av_wav: %s: unsupported format (tag %d, %d bits, %d channels, block %d): PCM 8/16/24/32 and IEEE float 32/64 are read", path,
                     w.tag, w.bits, w.channels, w.block);
        return AV_ERR_ARG;
    }
    return AV_OK;
}

}  // namespace

extern "C" int av_wav_info(const char* path, int* sample_rate, int* channels, long long* frames, int* bits, int* is_float) {
    AV_CHECK(path && sample_rate && channels && frames, "av_wav_info: null pointer");
    FILE* f = fopen(path, "rb");
    if (!f) { av_set_error("av_wav_info: cannot open %s", path); return AV_ERR_ARG; }
    WavFmt w;
    int rc = parse_wav(f, w, path);
    fclose(f);
    if (rc != AV_OK) return rc;
    if ((rc = check_fmt(w, path)) != AV_OK) return rc;
    *sample_rate = w.rate; *channels = w.channels; *frames = w.data_bytes / w.block;
    if (bits) *bits = w.bits;
    if (is_float) *is_float = w.tag == 3;
    return AV_OK;
}

// frames [frame0, frame0 + n) of the file -> out[n] float32 mono (HOST memory)
extern "C" int av_wav_read_mono_f32(const char* path, long long frame0, long long n, float* out) {
    AV_CHECK(path && (out || n == 0) && frame0 >= 0 && n >= 0, "av_wav_read_mono_f32: bad args");
    FILE* f = fopen(path, "rb");
    if (!f) { av_set_error("av_wav_read_mono_f32: cannot open %s", path); return AV_ERR_ARG; }
    WavFmt w;
    int rc = parse_wav(f, w, path);
    if (rc == AV_OK) rc = check_fmt(w, path);
    if (rc != AV_OK) { fclose(f); return rc; }
    const long long total = w.data_bytes / w.block;
    if (frame0 + n > total) { fclose(f); av_set_error("av_wav_read_mono_f32: %s holds %lld frames, [%lld, %lld) requested", path, total, frame0, frame0 + n); return AV_ERR_ARG; }
    const int C = w.channels, bs = w.bits / 8;
    const long long CHUNK = 1 << 16;
    std::vector<unsigned char> buf((size_t)(CHUNK * w.block));
    fseek(f, (long)(w.data_off + frame0 * w.block), SEEK_SET);
    const float invC = 1.0f / (float)C;
    for (long long i0 = 0; i0 < n; i0 += CHUNK) {
        const long long m = n - i0 < CHUNK ? n - i0 : CHUNK;
        if ((long long)fread(buf.data(), (size_t)w.block, (size_t)m, f) != m) { fclose(f); av_set_error("av_wav_read_mono_f32: %s: short read", path); return AV_ERR_ARG; }
        for (long long i = 0; i < m; ++i) {
            const unsigned char* p = buf.data() + i * w.block;
            float s = 0.f;
            for (int c = 0; c < C; ++c, p += bs) {
                float v;
                if (w.tag == 3) {
                    if (bs == 4) { memcpy(&v, p, 4); } else { double d; memcpy(&d, p, 8); v = (float)d; }
                } else if (bs == 1) v = ((float)p[0] - 128.f) * (1.0f / 128.f);
                else if (bs == 2) v = (float)(short)rd_u16(p) * (1.0f / 32768.f);
                else if (bs == 3) v = (float)((int)((unsigned)p[0] << 8 | (unsigned)p[1] << 16 | (unsigned)p[2] << 24) >> 8) * (1.0f / 8388608.f);
                else v = (float)(int)rd_u32(p) * (1.0f / 2147483648.f);
                s = c == 0 ? v : s + v;                              // float32 running sum, then / C  (np.mean over the channel axis)
            }
            out[i0 + i] = C == 1 ? s : s * invC;
        }
    }
    fclose(f);
    return AV_OK;
}
