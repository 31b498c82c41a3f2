/* synth.c — synthetic work-list generator (placeholder, filled in below) */
int m355_synth_version(void) { return 0; }
