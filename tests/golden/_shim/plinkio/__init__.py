"""Empty stand-in: only make_ld_schema touches plinkio."""
