// kernel translation unit 1 of 8: see acme_hip_part.inc
#define ACME_PART 1
#include "acme_hip_part.inc"
