// mid-size kernel, 20 columns in registers: see acme_hip_coop.inc
#define ACME_COOP_NC 20
#include "acme_hip_coop.inc"
