// kernel translation unit 2 of 8: see acme_hip_part.inc
#define ACME_PART 2
#include "acme_hip_part.inc"
