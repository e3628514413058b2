/*
 * smoke.c — include/alacgpu.h used from plain C99 (gcc -std=c99 -Wall -Wextra -pedantic), the way a cgo / FFI binding
 * sees it: no C++ and nothing of this repository but the header and the shared library.
 *
 *   smoke              decodes the hand-derived known-answer packets K1..K4 through
 *                      alacgpu_decode_packet and alacgpu_decode_batch and checks every byte (needs a GPU)
 *   smoke --link-only  only checks that every declared entry point links and that create() rejects bad configurations
 *                      before touching a device (the CPU test-suite runs this)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "alacgpu.h"

static int hexval(int c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; }

static size_t unhex(const char* s, uint8_t* out) {
    size_t n = 0;
    while (s[0] && s[1]) {
        out[n++] = (uint8_t)(hexval(s[0]) * 16 + hexval(s[1]));
        s += 2;
    }
    return n;
}

struct kat {
    const char* name;
    uint32_t frame_length;
    uint8_t channels;
    const char* packet;
    const char* pcm;
};

static const struct kat kats[] = {
    {"K1 mono escape", 4, 1, "0000020003FFFEFFFF0001C0", "0100FFFFFF7F0080"},
    {"K2 mono all-zero", 8, 1, "0000000000010047", "00000000000000000000000000000000"},
    {"K3 mono residuals", 4, 1, "0000000000010181CE", "0100FFFF02000000"},
    {"K4 stereo mix + escape code", 2, 2, "2000000402010001018EF7FC0017C0", "03000100FEFF0400"},
};

int main(int argc, char** argv) {
    const int link_only = argc > 1 && strcmp(argv[1], "--link-only") == 0;
    alacgpu_config cfg;
    alacgpu_decoder* dec = NULL;
    size_t k;

    memset(&cfg, 0, sizeof cfg);
    cfg.frame_length = 4096;
    cfg.bit_depth = 13; /* decoder.go:91-93: ErrConfig before anything else happens */
    cfg.num_channels = 2;
    if (alacgpu_create(&cfg, 0, &dec) != ALACGPU_E_CONFIG || dec != NULL) {
        fprintf(stderr, "bit depth 13 was not rejected: %s\n", alacgpu_last_error());
        return 1;
    }
    if (strstr(alacgpu_last_error(), "bit depth") == NULL) return 1;
    if (strstr(alacgpu_version(), "gfx950") == NULL) return 1;
    if (link_only) {
        /* take the address of every entry point so that the linker must resolve it */
        void (*fns[24])(void);
        size_t i = 0;
        fns[i++] = (void (*)(void))alacgpu_create;
        fns[i++] = (void (*)(void))alacgpu_destroy;
        fns[i++] = (void (*)(void))alacgpu_get_format;
        fns[i++] = (void (*)(void))alacgpu_frame_bytes;
        fns[i++] = (void (*)(void))alacgpu_decode_packet;
        fns[i++] = (void (*)(void))alacgpu_decode_batch;
        fns[i++] = (void (*)(void))alacgpu_decode_batch_device;
        fns[i++] = (void (*)(void))alacgpu_reserve;
        fns[i++] = (void (*)(void))alacgpu_last_kernel_ms;
        fns[i++] = (void (*)(void))alacgpu_timing_reset;
        fns[i++] = (void (*)(void))alacgpu_kernel_times;
        fns[i++] = (void (*)(void))alacgpu_stream;
        fns[i++] = (void (*)(void))alacgpu_synchronize;
        fns[i++] = (void (*)(void))alacgpu_last_error;
        fns[i++] = (void (*)(void))alacgpu_version;
        fns[i++] = (void (*)(void))alacgpu_trim;
        fns[i++] = (void (*)(void))alacgpu_pair_placement;
        fns[i++] = (void (*)(void))alacgpu_last_dispatch;
        fns[i++] = (void (*)(void))alacgpu_decode_batch_start;
        fns[i++] = (void (*)(void))alacgpu_decode_batch_wait;
        fns[i++] = (void (*)(void))alacgpu_host_alloc;
        fns[i++] = (void (*)(void))alacgpu_host_free;
        while (i--)
            if (fns[i] == NULL) return 1;
        printf("c_abi smoke: linked, %s\n", alacgpu_version());
        return 0;
    }

    for (k = 0; k < sizeof kats / sizeof kats[0]; k++) {
        uint8_t packet[64], want[64], out[64], bout[3 * 64];
        uint64_t offsets[4];
        uint32_t frames[3];
        int32_t status[3], st = -1;
        alacgpu_format fmt;
        size_t plen = unhex(kats[k].packet, packet), wlen = unhex(kats[k].pcm, want), n = 0, fb, i;
        uint8_t blob[3 * 64];

        memset(&cfg, 0, sizeof cfg);
        cfg.frame_length = kats[k].frame_length;
        cfg.bit_depth = 16;
        cfg.num_channels = kats[k].channels;
        cfg.pb = 40;
        cfg.mb = 10;
        cfg.kb = 14;
        cfg.max_run = 255;
        cfg.sample_rate = 44100;
        if (alacgpu_create(&cfg, 0, &dec) != ALACGPU_E_OK) {
            fprintf(stderr, "%s: create: %s\n", kats[k].name, alacgpu_last_error());
            return 1;
        }
        fb = alacgpu_frame_bytes(dec);
        if (fb != wlen || alacgpu_get_format(dec, &fmt) != ALACGPU_E_OK || fmt.channels != kats[k].channels ||
            fmt.bit_depth != 16 || fmt.sample_rate != 44100)
            return 1;
        /* DecodePacket (decoder.go:117) */
        if (alacgpu_decode_packet(dec, packet, plen, out, sizeof out, &n, &st) != ALACGPU_E_OK || st != 0 || n != wlen ||
            memcmp(out, want, wlen) != 0) {
            fprintf(stderr, "%s: decode_packet: status %#x, %lu bytes: %s\n", kats[k].name, (unsigned)st, (unsigned long)n,
                    alacgpu_last_error());
            return 1;
        }
        /* DecodePackets: the packet, a truncated copy whose neighbour is NOT zero bytes, the packet again, all dense */
        memcpy(blob, packet, plen);
        memcpy(blob + plen, packet, plen / 2);
        memcpy(blob + plen + plen / 2, packet, plen);
        offsets[0] = 0;
        offsets[1] = plen;
        offsets[2] = plen + plen / 2;
        offsets[3] = 2 * plen + plen / 2;
        if (alacgpu_decode_batch(dec, blob, (size_t)(2 * plen + plen / 2), offsets, 3, bout, fb, frames, status) != ALACGPU_E_OK) {
            fprintf(stderr, "%s: decode_batch: %s\n", kats[k].name, alacgpu_last_error());
            return 1;
        }
        for (i = 0; i < 3; i += 2)
            if (status[i] != 0 || frames[i] != kats[k].frame_length || memcmp(bout + i * fb, want, wlen) != 0) {
                fprintf(stderr, "%s: batch slot %lu: status %#x\n", kats[k].name, (unsigned long)i, (unsigned)status[i]);
                return 1;
            }
        if (status[1] == 0 && frames[1] == kats[k].frame_length && memcmp(bout + fb, want, wlen) == 0 && plen / 2 < plen - 2) {
            /* half a packet cannot decode to the whole answer unless it read its neighbour's bytes */
            fprintf(stderr, "%s: the truncated packet decoded as if it were whole\n", kats[k].name);
            return 1;
        }
        alacgpu_destroy(dec);
        dec = NULL;
        printf("c_abi smoke: %s ok\n", kats[k].name);
    }
    return 0;
}
