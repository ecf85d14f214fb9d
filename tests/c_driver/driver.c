/* Plain-C driver of the C ABI (include/mic_hip.h): what a cgo shim would do, without Go.
 * PICS-8 round trip of a synthetic 12-bit frame through host buffers, one batch call, one session call.
 * Build: gcc -O2 -I include tests/c_driver/driver.c -L medical-image-codec_amd -lmic_hip -Wl,-rpath,... -o driver
 * Exit code 0 = every call returned MIC_OK and the pixels came back bit-identical. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mic_hip.h"

static uint32_t rnd(uint32_t *s) { *s ^= *s << 13; *s ^= *s >> 17; *s ^= *s << 5; return *s; }

int main(void) {
    const int W = 640, H = 480, STRIPS = 8;
    const uint16_t MAXV = 4095;
    size_t npx = (size_t)W * H;
    uint16_t *img = malloc(npx * 2), *back = malloc(npx * 2);
    uint32_t seed = 12345;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            img[(size_t)y * W + x] = (uint16_t)((1000 + (x * 3 + y * 2) / 4 + (rnd(&seed) % 9)) & MAXV);
    if (mic_hip_set_device(0) != MIC_OK) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    printf("device: %s, library %s\n", mic_hip_device_name(), mic_hip_version());

    /* CompressParallelStrips / DecompressParallelStrips */
    size_t cap = MIC_HIP_PICS_BOUND(W, H, STRIPS), clen = 0;            /* the one capacity contract, include/mic_hip.h */
    uint8_t *comp = malloc(cap);
    int rc = mic_hip_pics_compress(img, W, H, MAXV, STRIPS, 2, comp, cap, &clen);
    if (rc != MIC_OK) { fprintf(stderr, "pics_compress rc=%d\n", rc); return 1; }
    int w = 0, h = 0, ns = 0, sh = 0;
    rc = mic_hip_pics_info(comp, clen, &w, &h, &ns, &sh);
    if (rc != MIC_OK || w != W || h != H || ns != STRIPS) { fprintf(stderr, "pics_info rc=%d %d %d %d\n", rc, w, h, ns); return 1; }
    memset(back, 0, npx * 2);
    rc = mic_hip_pics_decompress(comp, clen, back, W, H);
    if (rc != MIC_OK || memcmp(img, back, npx * 2) != 0) { fprintf(stderr, "pics_decompress rc=%d / mismatch\n", rc); return 1; }
    printf("PICS-%d: %zu -> %zu bytes (ratio %.3f), round trip ok\n", STRIPS, npx * 2, clen, (double)(npx * 2) / (double)clen);

    /* one batch call: four frames (the crossing a cgo shim would make once per slice) */
    enum { NJ = 4 };
    mic_hip_enc_job ej[NJ]; mic_hip_dec_job dj[NJ];
    uint8_t *outs[NJ];
    memset(ej, 0, sizeof ej); memset(dj, 0, sizeof dj);
    for (int i = 0; i < NJ; i++) {
        outs[i] = malloc(MIC_HIP_FRAME_BOUND(npx / NJ));
        ej[i].pixels = img + (size_t)i * (H / NJ) * W; ej[i].width = W; ej[i].height = H / NJ; ej[i].max_value = MAXV; ej[i].nstates = 2;
        ej[i].out = outs[i]; ej[i].out_cap = MIC_HIP_FRAME_BOUND(npx / NJ);
    }
    rc = mic_hip_compress_batch(ej, NJ);
    if (rc != MIC_OK) { fprintf(stderr, "compress_batch rc=%d\n", rc); return 1; }
    for (int i = 0; i < NJ; i++) {
        if (ej[i].status != MIC_OK) { fprintf(stderr, "job %d status %d\n", i, ej[i].status); return 1; }
        dj[i].compressed = outs[i]; dj[i].compressed_len = ej[i].out_len;
        dj[i].pixels_out = back + (size_t)i * (H / NJ) * W; dj[i].width = W; dj[i].height = H / NJ;
    }
    memset(back, 0, npx * 2);
    rc = mic_hip_decompress_batch(dj, NJ);
    if (rc != MIC_OK) { fprintf(stderr, "decompress_batch rc=%d\n", rc); return 1; }
    for (int i = 0; i < NJ; i++) if (dj[i].status != MIC_OK) { fprintf(stderr, "dec job %d status %d\n", i, dj[i].status); return 1; }
    if (memcmp(img, back, npx * 2) != 0) { fprintf(stderr, "batch mismatch\n"); return 1; }
    printf("batch of %d frames: round trip ok\n", NJ);

    /* round 4: two shards on one device (what a host with several GPUs does with their indices), the plan of the cut, and the
     * strip index a failing PICS call names (parallelstrips.go:97) */
    {
        const int two[2] = { 0, 0 };
        int got[4] = { -1, -1, -1, -1 };
        if (mic_hip_set_devices(two, 2) != MIC_OK || mic_hip_get_devices(got, 4) != 2 || got[0] != 0 || got[1] != 0) { fprintf(stderr, "set_devices\n"); return 1; }
        const uint64_t wts[5] = { 10, 10, 10, 10, 40 };
        int first[3] = { -1, -1, -1 };
        if (mic_hip_shard_plan(wts, 5, 2, first) != MIC_OK || first[0] != 0 || first[2] != 5 || first[1] < 1 || first[1] > 4) { fprintf(stderr, "shard_plan %d %d %d\n", first[0], first[1], first[2]); return 1; }
        memset(back, 0, npx * 2);
        rc = mic_hip_compress_batch(ej, NJ);
        for (int i = 0; rc == MIC_OK && i < NJ; i++) { if (ej[i].status != MIC_OK) rc = ej[i].status; dj[i].compressed_len = ej[i].out_len; }
        if (rc == MIC_OK) rc = mic_hip_decompress_batch(dj, NJ);
        if (rc != MIC_OK || memcmp(img, back, npx * 2) != 0) { fprintf(stderr, "sharded batch rc=%d / mismatch\n", rc); return 1; }
        const int one[1] = { 0 };
        if (mic_hip_set_devices(one, 1) != MIC_OK) return 1;
        /* strip 5 of 32 (15 rows, 9600 pixels) is uniform noise over the whole 12-bit range: too few pixels for their alphabet, no FSE
         * flavour makes it smaller -- the call fails and says where */
        enum { NS = 32 };
        const size_t cap32 = MIC_HIP_PICS_BOUND(W, H, NS);
        uint8_t *comp32 = malloc(cap32);
        uint16_t *bad = malloc(npx * 2);
        memcpy(bad, img, npx * 2);
        for (size_t i = (size_t)(5 * (H / NS)) * W; i < (size_t)(6 * (H / NS)) * W; i++) bad[i] = (uint16_t)(rnd(&seed) & MAXV);
        int strip = -7; size_t blen = 0;
        rc = mic_hip_pics_compress_ex(bad, W, H, MAXV, NS, 2, comp32, cap32, &blen, &strip);
        if (rc == MIC_OK || strip != 5) { fprintf(stderr, "pics_compress_ex rc=%d strip=%d (expected a failure in strip 5)\n", rc, strip); return 1; }
        strip = -7;
        rc = mic_hip_pics_compress_ex(img, W, H, MAXV, 4, 2, comp, cap, &blen, &strip);
        if (rc != MIC_OK || strip != -1) { fprintf(stderr, "pics_compress_ex on a good image rc=%d strip=%d\n", rc, strip); return 1; }
        comp[blen - 1] = 0;                                               /* the last strip loses its end mark (bitreader.go:36-38) */
        strip = -7;
        rc = mic_hip_pics_decompress_ex(comp, blen, back, W, H, &strip);
        if (rc == MIC_OK || strip != 3) { fprintf(stderr, "pics_decompress_ex rc=%d strip=%d (expected a failure in strip 3)\n", rc, strip); return 1; }
        free(comp32);
        free(bad);
        printf("two shards, shard plan, failing strips named: ok\n");
    }
    free(img); free(back); free(comp);
    for (int i = 0; i < NJ; i++) free(outs[i]);
    return 0;
}
