/* stb_dump -- the reference's image read, as a program: stbi_load(name, &w, &h, &n, 1) (Deff2DGPU/Deff2D.cuh:342, :377).
 * TEST INFRASTRUCTURE (oracle/_ref): built from the reference's own stb_image.h where it lies under /root/reference
 * (oracle/Makefile, target `ref`); writes the w*h gray bytes to argv[2] and prints "w h file_channels". */
#include <stdio.h>
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
int main(int argc, char **argv)
{
    int w = 0, h = 0, n = 0;
    if (argc < 3) { fprintf(stderr, "usage: stb_dump in.jpg out.raw\n"); return 2; }
    unsigned char *p = stbi_load(argv[1], &w, &h, &n, 1);
    if (!p) { fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason()); return 1; }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(p, 1, (size_t)w * h, f);
    fclose(f);
    printf("%d %d %d\n", w, h, n);
    return 0;
}
