"""sacrebleu stand-in (only imported during fairseq task auto-import)."""
