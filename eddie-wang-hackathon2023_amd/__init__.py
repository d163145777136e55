"""whisper-mi355: MI355X-native Whisper quantized-inference engine.

The directory is laid out like the reference's examples/whisper/ script directory: modules import
each other flatly (`from encoding import WhisperEncoding`), so put this directory on sys.path
(tests/conftest.py, bench.py and __graft_entry__.py do).
"""
