/* Exhaustive check: the oracle's sincosf restatement vs the host libm for every float in
 * [0, 6.2832] (1 086 918 650 values).  Build and run:
 *   gcc -O2 -ffp-contract=off -o /tmp/sweep tools/sincosf_sweep.c oracle/orb_oracle.c -Ioracle -lm && /tmp/sweep
 * Result in this container (glibc 2.35, x86-64): n=1086918650 bad=0. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
void orc_sincosf(float x, float *s, float *c);
int main(void)
{
    uint32_t hi;
    float top = 6.2832f;
    memcpy(&hi, &top, 4);
    long bad = 0, n = 0;
    for (uint32_t u = 0; u <= hi; u++) {
        float x, s, c, s2, c2;
        memcpy(&x, &u, 4);
        orc_sincosf(x, &s, &c);
        sincosf(x, &s2, &c2);
        n++;
        if (memcmp(&s, &s2, 4) || memcmp(&c, &c2, 4)) {
            if (bad < 10) printf("x=%a s=%a/%a c=%a/%a\n", x, s, s2, c, c2);
            bad++;
        }
    }
    printf("n=%ld bad=%ld\n", n, bad);
    return bad != 0;
}
