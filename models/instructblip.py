"""reference models/instructblip.py surface -> MI355X implementation."""
from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration, seed  # noqa: F401
