/* asan_check.c -- sanitizer run of the restatement (CPU only; SURVEY 5: the restatement must be
 * ASan/UBSan clean, unlike the reference which reads 1 byte OOB at cpu.h:517).  TEST INFRASTRUCTURE ONLY.
 * usage: asan_check file w h qp */
#include "deblock_oracle.h"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    dbko_frame *f = NULL;
    int rc = dbko_frame_create_from_file(&f, argv[1], (unsigned)atoi(argv[2]), (unsigned)atoi(argv[3]));
    if (rc) { fprintf(stderr, "create: %d\n", rc); return 1; }
    dbko_qp qp = { (unsigned)atoi(argv[4]), NULL, 0, 6 };
    rc = dbko_frame_filter(f, &qp, NULL, 7u, 2);
    dbko_frame_destroy(f);
    printf("asan_check ok rc=%d\n", rc);
    return rc;
}
