// placeholder until the chunk-parallel reverb lands
DEVO bool reverb_fast(PgFx&, float*, int, FastCtx&) { return false; }
