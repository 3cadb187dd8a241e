/*
 * ref_cli.c -- TEST INFRASTRUCTURE.  Runs the reference's own main()
 * (renamed ref_main by oracle/Makefile) so that a whole reference run,
 * argument parsing and file output included, can be started as a child
 * process: `ref_cli_f7 <lcg seed> size depth spp out [aperture]`.
 * The scene is the one reference PathTracer.c includes (buildScene.c).
 */
#include <stdint.h>
#include <stdlib.h>
int ref_main(int argc, char *argv[]);
void ref_rng_lcg(uint64_t seed);
int main(int argc, char *argv[]) {
    if (argc < 2) return 2;
    ref_rng_lcg(strtoull(argv[1], 0, 10));
    return ref_main(argc - 1, argv + 1);
}
