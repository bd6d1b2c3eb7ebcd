"""autoreparam_amd: MI355X-native HMC / mean-field VI engine for the hot path of
mgorinova/autoreparam."""
__version__ = "0.1.0"
