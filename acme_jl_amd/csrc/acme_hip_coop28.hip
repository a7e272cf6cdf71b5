// mid-size kernel, 28 columns in registers: see acme_hip_coop.inc
#define ACME_COOP_NC 28
#include "acme_hip_coop.inc"
