// main.cpp -- `star_amd`, the executable: the whole front end lives in cli_run.cpp (include/star_amd_cli.h) so that bench.py and the
// tests can run the identical pipeline in-process.
#include "../../../include/star_amd_cli.h"
int main(int argc, char **argv) { return staramd_cli_main(argc, argv, nullptr, nullptr); }
