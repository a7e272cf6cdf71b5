// mid-size kernel, 32 columns in registers: see acme_hip_coop.inc
#define ACME_COOP_NC 32
#include "acme_hip_coop.inc"
