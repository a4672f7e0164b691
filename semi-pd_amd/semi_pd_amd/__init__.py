"""semi_pd_amd — MI355X-native Semi-PD serving hot path (host side over libsemipd_hip.so)."""
__version__ = "0.1.0"
