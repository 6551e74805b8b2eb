"""MI355X-native Ladder-VAE hot path (import as `lvae_amd`, see /lvae_amd.py)."""
__version__ = '0.1.0'
