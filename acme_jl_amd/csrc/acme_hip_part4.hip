// kernel translation unit 4 of 8: see acme_hip_part.inc
#define ACME_PART 4
#include "acme_hip_part.inc"
