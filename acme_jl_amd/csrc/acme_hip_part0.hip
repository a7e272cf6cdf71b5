// kernel translation unit 0 of 8: see acme_hip_part.inc
#define ACME_PART 0
#include "acme_hip_part.inc"
