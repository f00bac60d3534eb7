"""empty stand-in so the reference modules import; fixture tooling only"""
def unidecode(s):
    return s
