"""MI355X-native audio-visual CTC path (package root; HIP library is loaded lazily by ``._lib``)."""
__all__ = []
