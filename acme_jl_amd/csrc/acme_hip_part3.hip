// kernel translation unit 3 of 8: see acme_hip_part.inc
#define ACME_PART 3
#include "acme_hip_part.inc"
