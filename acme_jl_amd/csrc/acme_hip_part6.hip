// kernel translation unit 6 of 8: see acme_hip_part.inc
#define ACME_PART 6
#include "acme_hip_part.inc"
