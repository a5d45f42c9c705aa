"""lit-parrot quantized decode path for MI355X (gfx950).  Import as ``lit_parrot_amd`` (see ../lit_parrot_amd)."""
