extern "C" const char* vd_source_sha(void) { return "b4bcfae032fb7a24"; }
