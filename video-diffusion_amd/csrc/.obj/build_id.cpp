extern "C" const char* vd_source_sha(void) { return "d22cb2d55434ea9e"; }
