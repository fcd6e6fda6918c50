"""Tunables of the simulation with the reference's defaults.

Mirrors the data side of the reference's `conf` singleton (src/conf.js:6-46): the
fields update() reads every frame and `updateParams()`.  The tweakpane GUI,
sensors and fps graph (src/conf.js:48-181) are browser-only and have no counterpart.
"""


class Conf:
    def __init__(self, mobile=False):
        # src/conf.js:8-30
        self.maxParticles = 8192 * 32
        self.particles = 8192 * 10
        self.run = True
        self.noise = 1.0
        self.gridNoiseStrength = 0.5
        self.speed = 1
        self.stiffness = 3.0
        self.restDensity = 1.0
        self.density = 1
        self.dynamicViscosity = 0.1
        self.gravity = 0
        self.actualSize = 1
        self.size = 0.5
        self.points = False
        self.mode = 0
        self.cursorInteraction = False
        self.hidePercentage = 0
        if mobile:                      # src/conf.js:33-36
            self.maxParticles = 8192 * 16
            self.particles = 4096
        self.updateParams()

    def updateParams(self):
        """src/conf.js:41-46."""
        level = max(self.particles / 8192, 1)
        size = 1.6 / pow(level, 1 / 3)
        self.actualSize = size * self.size
        self.restDensity = 0.25 * level * self.density


conf = Conf()
