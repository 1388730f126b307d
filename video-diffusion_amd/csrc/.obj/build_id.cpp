extern "C" const char* vd_source_sha(void) { return "bc1a360091d3f91f"; }
