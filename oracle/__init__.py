"""CPU oracle for libsfx: test infrastructure only (see oracle/sfx_oracle.h)."""
