// kernel translation unit 7 of 8: see acme_hip_part.inc
#define ACME_PART 7
#include "acme_hip_part.inc"
