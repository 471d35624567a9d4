"""CPU oracle -- TEST INFRASTRUCTURE ONLY.  See oracle/orc.h for the parity statement."""
