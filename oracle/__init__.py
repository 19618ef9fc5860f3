"""TEST INFRASTRUCTURE ONLY -- see oracle/hamming_map.py."""
