// kernel translation unit 2 of 4: see acme_hip_part.inc
#define ACME_PART 2
#include "acme_hip_part.inc"
