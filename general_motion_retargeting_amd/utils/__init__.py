"""Source adapters (the callers' side of the hot path): SURVEY.md section 8(f)."""
