"""empty stand-in so the reference modules import; fixture tooling only"""
class Generator(object):
    pass
