// kernel translation unit 5 of 8: see acme_hip_part.inc
#define ACME_PART 5
#include "acme_hip_part.inc"
