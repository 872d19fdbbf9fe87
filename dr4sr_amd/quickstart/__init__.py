from .run import run
