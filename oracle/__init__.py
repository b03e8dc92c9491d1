"""ORACLE package -- CPU restatements used only as the checker (tests/, smoke(), bench.py cpu_baseline).
Nothing under ``cartnet_amd`` may import from here."""
