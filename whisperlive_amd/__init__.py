"""whisperlive_amd — MI355X-native (gfx950) Whisper streaming-inference hot path behind collabora/WhisperLive's
backend contract. Importing the package never touches the GPU; the HIP library is loaded on first engine use and
there is no CPU fallback (whisperlive_amd._lib.load raises if libwlx.so is missing)."""
__version__ = "0.1.0"
