// kernel translation unit 3 of 4: see acme_hip_part.inc
#define ACME_PART 3
#include "acme_hip_part.inc"
