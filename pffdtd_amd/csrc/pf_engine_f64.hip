// pf_engine_f64.hip -- the fp64 instantiation of the engine (pf_engine_class.inc), a translation unit of its own so that it compiles beside
// the fp32 one (pf_engine.hip, which also holds the C ABI).
#include "pf_engine_class.inc"

pfeng::EngineBase *pf__new_engine_f64(const pf_simdata *sd, const pf_opts_x *o, int *rc) {
   auto *e = new Engine<double>();
   *rc = e->init(sd, o);
   return e;
}
