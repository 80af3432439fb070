/* Plain-C use of libfacet_engine.so (include/facet_engine.h): what a non-Python host would do to score a batch.
 *
 *   gcc -std=c99 -Iinclude examples/score_topiq.c -o /tmp/score_topiq -Lfacet_amd -lfacet_engine -Wl,-rpath,$PWD/facet_amd
 *   /tmp/score_topiq [model.onnx]
 *
 * Without a gfx950 device fe_create fails with a message (there is no CPU fallback) and the program says so and exits 0 after
 * exercising the host-only entry points; with a device it loads a weight set the caller provides through fe_weights_set (here:
 * none, so fe_topiq_score reports FE_ERR_NOT_LOADED) - the point of the example is the calling convention, not a checkpoint. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "facet_engine.h"

int main(int argc, char** argv) {
  printf("engine: %s\n", fe_version());
  if (argc > 1) {                       /* host-only: validate an .onnx file before any device is involved */
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* buf = malloc((size_t)len);
    if (fread(buf, 1, (size_t)len, f) != (size_t)len) { fclose(f); return 2; }
    fclose(f);
    int nodes = 0, inits = 0, outs = 0;
    int64_t dims[4];
    char err[256] = "";
    if (fe_onnx_probe(buf, (size_t)len, &nodes, &inits, &outs, dims, err, (int)sizeof err) == FE_OK)
      printf("%s: %d nodes, %d initializers, %d outputs, input [%lld,%lld,%lld,%lld]\n", argv[1], nodes, inits, outs, (long long)dims[0],
             (long long)dims[1], (long long)dims[2], (long long)dims[3]);
    else
      printf("%s rejected: %s\n", argv[1], err);
    free(buf);
  }
  fe_ctx* ctx = NULL;
  if (fe_create(0, 0, &ctx) != FE_OK) {
    printf("no engine context: %s\n", fe_last_error(NULL));
    return 0;
  }
  enum { N = 2, H = 64, W = 96 };
  unsigned char* rgb = (unsigned char*)calloc((size_t)N * H * W * 3, 1);
  float scores[N];
  int rc = fe_topiq_score(ctx, rgb, N, H, W, /*on_device=*/0, scores);
  if (rc != FE_OK) printf("fe_topiq_score -> %d: %s\n", rc, fe_last_error(ctx));   /* weights were never loaded */
  else printf("scores: %f %f\n", scores[0], scores[1]);
  free(rgb);
  fe_destroy(ctx);
  return 0;
}
