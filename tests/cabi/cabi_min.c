/* TEST TOOL: the C ABI from plain C99 (gcc -std=c99 -pedantic): include/pddp.h must not need a C++ compiler. */
#include <stdio.h>
#include "pddp.h"
int main(void) {
    pddp_config c;
    if (pddp_default_config(&c, 4)) { fprintf(stderr, "%s\n", pddp_last_error()); return 1; }
    c.N = 32; c.A = 4; c.M = 4; c.batch = 1;
    pddp_handle h = 0;
    int rc = pddp_create(&c, &h);
    printf("create rc %d (%s)\n", rc, rc ? pddp_last_error() : "ok");
    if (!rc) pddp_destroy(h);
    return 0;
}
