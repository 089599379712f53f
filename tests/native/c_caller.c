/* c_caller.c — a plain C99 caller of include/drfe.h (no C++, no Python): create, extract one frame, destroy.
 * usage: c_caller <gray.raw> <w> <h> <out.bin>  -> out.bin = int32 n, n keypoints (28 B), n descriptors (32 B) */
#include "drfe.h"

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(gray, 1, (size_t)w * h, f) != (size_t)w * h) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    drfe_config cfg = {0, w, h, 1, 1000, 1.2f, 8, 20, 7};
    drfe_ctx* ctx = NULL;
    if (drfe_create(&cfg, &ctx) != DRFE_OK) { fprintf(stderr, "drfe_create: %s\n", drfe_last_error(NULL)); return 1; }
    const int cap = drfe_orb_max_keypoints(ctx);
    drfe_keypoint* kps = (drfe_keypoint*)malloc(sizeof(drfe_keypoint) * (size_t)cap);
    uint8_t* desc = (uint8_t*)malloc(32 * (size_t)cap);
    int n = 0;
    int rc = drfe_orb_extract(ctx, gray, w, h, (size_t)w, kps, desc, cap, &n);
    if (rc != DRFE_OK) { fprintf(stderr, "drfe_orb_extract: %s\n", drfe_last_error(ctx)); return 1; }
    /* error behaviour of the boundary: too small a buffer is refused, nothing partial is written */
    if (drfe_orb_extract(ctx, gray, w, h, (size_t)w, kps, desc, 1, &rc) != DRFE_ERR_CAPACITY) return 3;
    if (drfe_orb_extract(ctx, NULL, 0, 0, 0, kps, desc, cap, &rc) != DRFE_OK || rc != 0) return 4;   /* empty image: silent */
    f = fopen(argv[4], "wb");
    int32_t n32 = n;
    fwrite(&n32, 4, 1, f);
    fwrite(kps, sizeof(drfe_keypoint), (size_t)n, f);
    fwrite(desc, 32, (size_t)n, f);
    fclose(f);
    printf("c_caller ok: %s, %d keypoints\n", drfe_version(), n);
    drfe_destroy(ctx);
    free(kps); free(desc); free(gray);
    return 0;
}
