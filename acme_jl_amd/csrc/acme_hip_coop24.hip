// mid-size kernel, 24 columns in registers: see acme_hip_coop.inc
#define ACME_COOP_NC 24
#include "acme_hip_coop.inc"
