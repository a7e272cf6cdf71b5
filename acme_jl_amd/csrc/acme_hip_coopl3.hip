// mid-size kernel, threshold path on a matrix in LDS, 3 rows per lane: see acme_hip_coop.inc
#define ACME_COOP_LDS_ROWS 3
#include "acme_hip_coop.inc"
