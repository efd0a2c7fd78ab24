/* oracle/sanitize/replay.c -- TEST INFRASTRUCTURE (as everything under oracle/): replays fuzz cases through the C restatement
 * (oracle/c/mmw_oracle.c) under the CPU sanitizers, to rule the CHECKER out when a GPU parity test fails once and passes on the
 * re-run (profiles/NOTEBOOK.md round 5: three such mismatches in ~85 000 fuzz runs under six processes).
 *   python oracle/sanitize/dump_cases.py            # writes /tmp/msan/*.bin: config bytes + points / counts / dt of the cases
 *   /opt/rocm/lib/llvm/bin/clang -fsanitize=memory -fsanitize-memory-track-origins=2 -g -O1 -Ioracle/c oracle/sanitize/replay.c \
 *       oracle/c/mmw_oracle.c -lm -o /tmp/msan/replay_msan && /tmp/msan/replay_msan /tmp/msan/*.bin
 *   gcc -fsanitize=address,undefined -g -O1 -Ioracle/c oracle/sanitize/replay.c oracle/c/mmw_oracle.c -lm -o /tmp/msan/replay_asan
 * Every byte of every track record, label vector and feature tensor the oracle hands out goes into a checksum that is branched
 * on, so MemorySanitizer reports any byte that was never written.  Result on 43 cases (the three failing seeds, 20 non-finite,
 * 20 regular): no report from MSan, ASan or UBSan. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "mmw_oracle.h"
int main(int argc, char **argv)
{
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        int hdr[4];
        if (!f || fread(hdr, 4, 4, f) != 4) { printf("bad %s\n", argv[a]); return 1; }
        const int csz = hdr[0], S = hdr[1], N = hdr[2], F = hdr[3];
        orc_config cfg;
        if (csz != (int)sizeof(cfg)) { printf("config size %d != %zu\n", csz, sizeof(cfg)); return 1; }
        fread(&cfg, 1, sizeof(cfg), f);
        double *pts = malloc((size_t)F * S * N * 8 * 8); int32_t *cnt = malloc((size_t)F * S * 4); double *dts = malloc((size_t)F * S * 8);
        fread(pts, 8, (size_t)F * S * N * 8, f); fread(cnt, 4, (size_t)F * S, f); fread(dts, 8, (size_t)F * S, f); fclose(f);
        orc_scene **sc = malloc(S * sizeof(*sc));
        for (int s = 0; s < S; s++) sc[s] = orc_scene_new(&cfg, N);
        const int ring = cfg.fb_frames_batch + 1;
        int32_t *assoc = malloc((size_t)N * 4), *lab = malloc((size_t)ring * N * 4 + 64);
        orc_track_record *rec = malloc(64 * sizeof(*rec));
        float *feat = malloc((size_t)64 * ring * 8 * 8 * 5 * 4); int32_t *owner = malloc(64 * 4);
        unsigned long long sum = 0; int errs = 0;
        for (int fr = 0; fr < F; fr++)
            for (int s = 0; s < S; s++) {
                const int c = cnt[fr * S + s];
                if (c == 0) continue;
                int32_t dbn = -1;
                const int rc = orc_track_frame(sc[s], pts + ((size_t)(fr * S + s) * N) * 8, c < 0 ? 0 : c, dts[fr * S + s], assoc, lab, &dbn);
                if (rc < 0 && rc > -7) { errs++; orc_scene_reset(sc[s]); continue; }
                for (int i = 0; i < (c < 0 ? 0 : c); i++) sum += (unsigned)assoc[i];
                if (dbn > 0) for (int i = 0; i < dbn; i++) sum += (unsigned)lab[i];
                const int nt = orc_get_tracks(sc[s], rec, 64);
                for (int t = 0; t < nt && t < 64; t++) {
                    const unsigned char *b = (const unsigned char *)&rec[t];
                    for (size_t k = 0; k < sizeof(rec[t]); k++) sum += b[k];   /* MSan: every byte of a record must be defined */
                }
                const int nf = orc_features(sc[s], feat, owner);
                for (int i = 0; i < nf * ring * 8 * 8 * 5; i++) { uint32_t u; memcpy(&u, feat + i, 4); sum += u; }
            }
        if (sum & 1) fputs("", stdout);   /* a branch on the checksum: MSan reports here if any byte that went into it was never written */
        printf("%s: S %d N %d F %d errors %d checksum %llx\n", argv[a], S, N, F, errs, sum);
        for (int s = 0; s < S; s++) orc_scene_free(sc[s]);
    }
    return 0;
}
