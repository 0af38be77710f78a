from .io import main
