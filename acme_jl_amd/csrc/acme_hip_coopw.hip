// mid-size kernel, threshold path on a matrix in LDS, one instance per wave (one row per lane): see acme_hip_coop.inc
#define ACME_COOP_WAVE 1
#include "acme_hip_coop.inc"
