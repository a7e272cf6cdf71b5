// mid-size kernel, threshold path on a matrix in LDS, 1 row per lane: see acme_hip_coop.inc
#define ACME_COOP_LDS_ROWS 1
#include "acme_hip_coop.inc"
