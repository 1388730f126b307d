extern "C" const char* vd_source_sha(void) { return "665eb5580451b407"; }
